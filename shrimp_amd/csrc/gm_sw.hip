// gm_sw.hip -- Smith-Waterman kernels: wave-level anti-diagonal DP (no MFMA: max-plus recurrence).
//
//   sw_vector_wave   score-only affine local SW, packed int16 (2 read rows per lane)
//                    ref: common/sw-vector.c:228-377 (vect_sw_same_gap / _diff_gap), :453-515
//   k_pass1          read_pass1_per_strand + f1_run           ref: gmapper/mapping.c:1261-1339, common/f1-wrapper.h:97-134
//   k_select         read_get_vector_hits (top-K ext-heap)    ref: gmapper/mapping.c:1376-1411, common/heap.h:226-327
//   k_pass2          hit_run_full_sw + sw_full_ls             ref: gmapper/mapping.c:331-402, common/sw-full-ls.c:154-516
//   k_sw_vector_batch  S1 batch entry
//
// sw_vector semantics.  The SSE2 code sweeps 8-row stripes along anti-diagonals with -1/-2
// sentinels around both sequences; the value it returns is exactly
//   H(i,j) = max(0, H(i-1,j-1)+s(i,j), A(i,j), B(i,j)),   s = match if codes equal else mismatch
//   A(i,j) = max(A(i,j-1) - a_ext, H(i,j-1) - a_open - a_ext)     (gap along the genome)
//   B(i,j) = max(B(i-1,j) - b_ext, H(i-1,j) - b_open - b_ext)     (gap along the read)
// maximised over the glen x rlen matrix, in int16 (no overflow: match*rlen < 32768, sw-vector.c:393).
// Pad cells (sentinel codes never match) cannot raise the maximum.  A wave holds read rows 2l and
// 2l+1 in the two halves of lane l's registers and advances one anti-diagonal per step; row r works
// on column t-r.  What row r needs from row r-1 arrives through one DPP wave shift + v_alignbit.
#include "gm_common.h"
#include "gm_internal.h"

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define SW_DB_SENT 0xF0u      // genome pad code (never equals a read code)
#define SW_QR_SENT 0xF1u      // read pad code

__device__ __forceinline__ uint32_t pk(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ s16x2 as_s(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_max(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u(as_s(a) - as_s(b)); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u(as_s(a) + as_s(b)); }

// min(x, 1) and a * b + c on both halves: one VOP3P instruction each (the generic vector builtins expand to compares and selects)
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t pk_mad_i16(uint32_t a, uint32_t b_uniform, uint32_t c) { uint32_t r; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c)); return r; }
// value of lane l-1, lane 0 gets 0 (bound_ctrl): no register to preset
__device__ __forceinline__ uint32_t wave_shr1_z(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
// value of lane l-1 (lane 0 keeps `lane0`): v_mov_b32_dpp wave_shr:1
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t lane0) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138, 0xf, 0xf, false);
}
// {low: prev-lane high half, high: own low half}
__device__ __forceinline__ uint32_t up_of(uint32_t own, uint32_t shifted) { return __builtin_amdgcn_alignbit(own, shifted, 16); }

// Score-only SW of db[0..glen) x qr[0..rlen) by one wave.  db/qr are byte arrays of 4-bit codes in
// LDS; carry is 2*(glen) int16 in LDS, used only when rlen > 128.  Every lane returns the score.
// CS: colour space -- read row 0 (the first colour) is compared with db0[c] = lstocs(genome_ls[c], initbp) instead of the colour
// genome (ref: common/sw-vector.c:112-146); the colour codes still flow on to row 1.
// SINGLE: rlen <= 128, one stripe -- no carry rows, lane 0's H neighbour is the zero row (the usual case: reads up to 128 bases).
// early_thr > 0 (pass 1 of unpaired reads only, where a window below the threshold is dropped and its score never read again): once the genome has
// run out (step t >= glen, the 41 % of the steps in which the anti-diagonal drains), the sweep stops as soon as NO alignment can reach early_thr any more.
// Every alignment that ends on a later anti-diagonal either passes through a cell of the last two anti-diagonals -- and gains at most match per remaining
// row / column after it: H(r, c) + match * min(rows - 1 - r, glen - 1 - c) -- or starts after them, which the same expression with H = 0 covers.  The value
// returned is then the maximum so far (< early_thr, like the final one) and *bounded is set; the caller keeps the two apart (f1 cache).
template <bool CS, bool SINGLE>
__device__ int sw_vector_wave_s(const uint8_t* db, const uint8_t* db0, int glen, const uint8_t* qr, int rlen, const GmScoreDev& sc,
                                int16_t* carry, int lane, const int early_thr = 0, bool* bounded = nullptr) {
  const uint32_t v_match = pk(sc.match, sc.match);
  const uint32_t v_delta = pk(sc.mismatch - sc.match, sc.mismatch - sc.match);
  const uint32_t v_a_ext = pk(sc.a_ge, sc.a_ge), v_a_oe = pk(sc.a_go + sc.a_ge, sc.a_go + sc.a_ge);
  const uint32_t v_b_ext = pk(sc.b_ge, sc.b_ge), v_b_oe = pk(sc.b_go + sc.b_ge, sc.b_go + sc.b_ge);
  const uint32_t v_one = pk(1, 1);
  uint32_t v_score = 0;
  const int n_stripes = SINGLE ? 1 : ((rlen + 127) >> 7);
  int16_t* carryH = carry; int16_t* carryB = carry + glen;
  // The last row of a stripe (H and the travelling gap value T, 16 bits each) for the stripe below: windows of up to 256 columns keep it in four registers -- column c in lane
  // c & 63 of register c >> 6, written from lane 63's value by a lane-select -- and need no LDS for it (a wave of pass 1 then holds 450 B instead of 1.3 KB at 150 bases, which
  // decides how many of them fit beside the seed lookup's tables); wider windows use the LDS rows.
  const bool creg = !SINGLE && glen <= 256;
  uint32_t cr0 = 0, cr1 = 0, cr2 = 0, cr3 = 0;
  auto cr_at = [&](const int k) -> uint32_t { return k == 0 ? cr0 : (k == 1 ? cr1 : (k == 2 ? cr2 : cr3)); };
  for (int s = 0; s < n_stripes; s++) {
    const int r0 = s * 128 + 2 * lane;
    const uint32_t q = pk(r0 < rlen ? qr[r0] : SW_QR_SENT, r0 + 1 < rlen ? qr[r0 + 1] : SW_QR_SENT);
    const int rows = min(128, rlen - s * 128);
    const int steps = glen + rows - 1;
    // Tprev: what the row below takes for its gap-along-the-read state, max(B - b_ext, H - b_open - b_ext) of this lane's rows at the column they just
    // left -- formed here, so that one value travels down a row per step instead of B and H both (two instructions less per step)
    uint32_t Hprev = 0, Aprev = pk(-sc.a_go, -sc.a_go), Tprev = pk(-sc.b_go - sc.b_ge, -sc.b_go - sc.b_ge);
    uint32_t Gprev = pk(SW_DB_SENT, SW_DB_SENT);
    uint32_t upH_prev = 0;                     // H(r-1, c-1) for the step to come
    const bool more = !SINGLE && (s + 1 < n_stripes);
    uint32_t dbr = 0, chv = 0, cbv = 0, db0v = 0;
    // The genome letter that enters at lane 0: the staging register of the next 64 columns (letter << 16) is rotated by one lane per step, so that lane 0
    // always holds the letter of the column it is about to start, and the shift of G takes it from there as the value its lane 0 keeps -- two DPP moves per
    // step instead of a lane read, a scalar shift, a move into a vector register and the DPP move.
    auto step = [&](const int t) {
      if ((t & 63) == 0) {                     // refill the per-lane staging of the next 64 columns
        const int c = t + lane;
        dbr = ((c < glen) ? (uint32_t)db[c] : SW_DB_SENT) << 16;
        if (CS && s == 0) db0v = (c < glen) ? (uint32_t)db0[c] : SW_DB_SENT;
        if (!SINGLE && s > 0) {
          if (creg) { const uint32_t w = cr_at(t >> 6); chv = (c < glen) ? (w & 0xFFFFu) : 0u; cbv = (c < glen) ? (w >> 16) : (uint32_t)(uint16_t)(-sc.b_go - sc.b_ge); }
          else { chv = (c < glen) ? (uint32_t)(uint16_t)carryH[c] : 0u; cbv = (c < glen) ? (uint32_t)(uint16_t)carryB[c] : (uint32_t)(uint16_t)(-sc.b_go - sc.b_ge); }
        }
      }
      const int sl = t & 63;
      // lane 0's neighbour (row 128s - 1) comes from the carry arrays / the initial row
      const uint32_t in_h = (!SINGLE && s > 0) ? ((uint32_t)__builtin_amdgcn_readlane((int)chv, sl) << 16) : 0u;
      const uint32_t in_b = (!SINGLE && s > 0) ? ((uint32_t)__builtin_amdgcn_readlane((int)cbv, sl) << 16) : ((uint32_t)(uint16_t)(-sc.b_go - sc.b_ge) << 16);   // (carryB holds T)
      const uint32_t dbn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dbr, 0x134, 0xf, 0xf, true);      // wave_rol:1 -- lane l takes lane l + 1's letter
      const uint32_t G = up_of(Gprev, (uint32_t)__builtin_amdgcn_update_dpp((int)dbr, (int)Gprev, 0x138, 0xf, 0xf, false));   // wave_shr:1, lane 0 keeps dbr's
      dbr = dbn;
      const uint32_t upH = up_of(Hprev, SINGLE ? wave_shr1_z(Hprev) : wave_shr1(Hprev, in_h));
      // SINGLE: row -1's b may be anything <= 0 -- a b value that is not positive never changes an H (H >= 0), and starting from 0 instead of
      // -b_open - b_ext the column of b stays <= 0 until an H - b_open - b_ext term takes over, which is the same term as in the exact recurrence.
      // a: gap along the genome (from the left), b: gap along the read (from above) = the T of the row above at this column
      const uint32_t b = up_of(Tprev, SINGLE ? wave_shr1_z(Tprev) : wave_shr1(Tprev, in_b));
      const uint32_t a = pk_max(pk_sub(Aprev, v_a_ext), pk_sub(Hprev, v_a_oe));
      uint32_t Gc = G;
      if (CS && s == 0) {                      // row 0 lives in the low half of lane 0
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)db0v, sl);
        if (lane == 0) Gc = (G & 0xFFFF0000u) | g0;
      }
      // s = match where the codes are equal, else mismatch: H(r-1, c-1) + min(code_xor, 1) * delta + match
      uint32_t h = pk_add(pk_mad_i16(pk_min_u16(Gc ^ q, v_one), v_delta, upH_prev), v_match);
      h = pk_max(pk_max(h, 0u), pk_max(a, b));
      v_score = pk_max(v_score, h);
      if (more && creg) {                      // row 128s+127 feeds the next stripe: lane 63's upper halves, into lane c & 63 of register c >> 6
        const int c = t - 127;
        if (c >= 0 && c < glen) {
          const uint32_t tn = pk_max(pk_sub(b, v_b_ext), pk_sub(h, v_b_oe));
          const uint32_t w = ((uint32_t)__builtin_amdgcn_readlane((int)h, 63) >> 16) | ((uint32_t)__builtin_amdgcn_readlane((int)tn, 63) & 0xFFFF0000u);
          const int k = c >> 6, ln = c & 63;
          const bool me = lane == ln;
          if (k == 0) cr0 = me ? w : cr0; else if (k == 1) cr1 = me ? w : cr1; else if (k == 2) cr2 = me ? w : cr2; else cr3 = me ? w : cr3;
        }
      } else
      if (more && lane == 63) {                // ... wider windows: the LDS rows
        const int c = t - 127;
        if (c >= 0 && c < glen) { carryH[c] = (int16_t)(h >> 16); carryB[c] = (int16_t)(pk_max(pk_sub(b, v_b_ext), pk_sub(h, v_b_oe)) >> 16); }
      }
      upH_prev = upH; Hprev = h; Aprev = a; Tprev = pk_max(pk_sub(b, v_b_ext), pk_sub(h, v_b_oe)); Gprev = G;
    };
    int t = 0;
    bool cut = false;
    // the most an alignment can still reach that crosses this stripe's last row in one of the columns 0 .. limit - 1, or starts below it
    auto carry_top = [&](const int limit) -> int {
      const int rows_left = rlen - (s + 1) * 128;
      int top = sc.match * min(rows_left, glen);
      for (int c = lane; c < limit; c += GM_WAVE) top = max(top, (creg ? (int)(int16_t)(cr_at(c >> 6) & 0xFFFFu) : (int)carryH[c]) + sc.match * min(rows_left, glen - 1 - c));
      for (int d = 32; d > 0; d >>= 1) top = max(top, __shfl_xor(top, d));
      return top;
    };
    if (early_thr > 0) {
      // (a stripe of a longer read: the rows of the stripes below count as rows left, and a stop here skips those stripes too)
      const int rows_below = SINGLE ? 0 : max(rlen - (s + 1) * 128, 0);
      const uint32_t v_row = pk(2 * lane, 2 * lane + 1);
      const uint32_t v_rleft = pk(max(rows - 1 - 2 * lane, 0) + rows_below, max(rows - 2 - 2 * lane, 0) + rows_below);
      const uint32_t v_thr1 = pk(early_thr - 1, early_thr - 1);
      for (int j = 0; j < 9; j++) {
        const int frac = (j == 0 ? 23 : j == 1 ? 33 : j == 2 ? 44 : j == 3 ? 56 : j == 4 ? 72 : j == 5 ? 92 : j == 6 ? 118 : j == 7 ? 154 : 192);     // of 256: where along the drain the test is made
        const int tc = (glen + ((rows * frac) >> 8) + 1) & ~1;
        if (tc + 8 >= steps) break;
        for (; t + 1 < tc; t += 2) { step(t); step(t + 1); }
        // steps 0 .. t - 1 are done: Hprev holds anti-diagonal t - 1 (row r at column t - 1 - r: r - (t - glen) columns after it), upH_prev anti-diagonal t - 2 one row down
        const int kk = t - glen;
        const uint32_t left = pk_min_u16(as_u(__builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, v_row), __builtin_bit_cast(u16x2, pk(kk, kk))))), v_rleft);
        const uint32_t pot = as_u(__builtin_bit_cast(s16x2, __builtin_bit_cast(u16x2, left) * __builtin_bit_cast(u16x2, v_match)));
        const uint32_t top = pk_max(v_score, pk_add(pk_max(Hprev, pk_add(upH_prev, v_match)), pot));
        const uint32_t over = as_u(__builtin_bit_cast(s16x2, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, top), __builtin_bit_cast(u16x2, v_thr1))));
        if (!__any(over != 0u)) {
          // (a stripe with stripes below it: an alignment may have LEFT this stripe already, through a cell of its last row that was computed before the
          // anti-diagonals tested above -- those cells are in the carry row: columns 0 .. t - 128)
          if (!SINGLE && more && carry_top(min(glen, t - 127)) >= early_thr) continue;
          cut = true; break;
        }
      }
    }
    if (!cut) {
      for (; t + 1 < steps; t += 2) { step(t); step(t + 1); }      // two steps per round (the second never starts a staging block: blocks begin at even t)
      if (t < steps) step(t);
    }
    if (bounded) *bounded = cut;
    if (more) __syncthreads();
    if (!SINGLE && cut) break;
    if (!SINGLE && more && early_thr > 0) {
      // Two stripes (reads of more than 128 bases): every alignment that ends in a later stripe crosses this stripe's last row in some cell (R, c) -- or starts behind it,
      // which H = 0 covers -- and gains at most `match` per row and column left; the gap state that crosses the row is below the cell's H.  If neither the best score so
      // far nor any H(R, c) + match * min(rows left, columns left) reaches the threshold, the remaining stripes (41 % of a 150-base read's steps) cannot either.
      int top = max((int)(int16_t)(v_score & 0xFFFF), (int)(int16_t)(v_score >> 16));
      for (int d = 32; d > 0; d >>= 1) top = max(top, __shfl_xor(top, d));
      if (max(top, carry_top(glen)) < early_thr) { if (bounded) *bounded = true; break; }
    }
  }
  int best = max((int)(int16_t)(v_score & 0xFFFF), (int)(int16_t)(v_score >> 16));
  for (int d = 32; d > 0; d >>= 1) best = max(best, __shfl_xor(best, d));
  return best;
}
template <bool CS>
__device__ __forceinline__ int sw_vector_wave_t(const uint8_t* db, const uint8_t* db0, int glen, const uint8_t* qr, int rlen, const GmScoreDev& sc,
                                                int16_t* carry, int lane, const int early_thr = 0, bool* bounded = nullptr) {
  if (bounded) *bounded = false;
  return rlen <= 128 ? sw_vector_wave_s<CS, true>(db, db0, glen, qr, rlen, sc, carry, lane, early_thr, bounded) : sw_vector_wave_s<CS, false>(db, db0, glen, qr, rlen, sc, carry, lane, early_thr, bounded);
}
__device__ int sw_vector_wave(const uint8_t* db, int glen, const uint8_t* qr, int rlen, const GmScoreDev& sc, int16_t* carry, int lane) {
  return sw_vector_wave_t<false>(db, nullptr, glen, qr, rlen, sc, carry, lane);
}

// unpack `len` codes starting at global position g0 into dst (forward) or reverse-complemented
__device__ void load_window(const uint32_t* __restrict__ genome, uint64_t g0, int len, bool rc, uint8_t* dst, int lane, bool rna = false) {      // rna: the CONTIG's flag
  const uint64_t w0 = g0 >> 3; const int sh = (int)(g0 & 7);
  const int nwords = (sh + len + 7) >> 3;
  const uint64_t cm = gm_cmpl_tab(rna);   // complement_base as nibbles (ref: util.h:125-151)
  for (int k = lane; k < nwords; k += GM_WAVE) {
    const uint32_t w = genome[w0 + k];
#pragma unroll
    for (int n = 0; n < 8; n++) {
      const int idx = k * 8 + n - sh;
      if (idx >= 0 && idx < len) {
        uint32_t c = (w >> (4 * n)) & 0xf;
        if (rc) { c = (uint32_t)(cm >> (c * 4)) & 0xf; dst[len - 1 - idx] = (uint8_t)c; }
        else dst[idx] = (uint8_t)c;
      }
    }
  }
}

__device__ void load_read(const uint32_t* __restrict__ rw, int read_len, bool rc, uint8_t* dst, int lane, bool rna = false) {      // rna: the READ's flag
  const uint64_t cm = gm_cmpl_tab(rna);
  for (int i = lane; i < read_len; i += GM_WAVE) {
    const int src = rc ? (read_len - 1 - i) : i;
    uint32_t c = (rw[src >> 3] >> ((src & 7) * 4)) & 0xf;
    if (rc) c = (uint32_t)(cm >> (c * 4)) & 0xf;
    dst[i] = (uint8_t)c;
  }
}

// sw_gapless (ref: common/sw-gapless.c:57-117) by one wave, letter space: the best ungapped segment on the contig diagonal through
// (g_idx, r_idx).  The reference's running score with reset below zero is the maximum-subarray sum: max over r of
// P[r] - min(0, P[j] for j < r) on the prefix sums P of the per-base scores -- two wave scans per 64 cells.
__device__ int sw_gapless_wave(const uint32_t* __restrict__ genome, uint64_t cbase, long long clen, const uint8_t* qr, int rlen,
                               long long g_idx, int r_idx, const GmScoreDev& sc, int lane) {
  long long g_left; int r_left;
  if (g_idx < r_idx) { g_left = 0; r_left = (int)(r_idx - g_idx); } else { g_left = g_idx - r_idx; r_left = 0; }
  const long long room = clen - g_left;
  const int n = (int)(room < (long long)(rlen - r_left) ? room : (long long)(rlen - r_left));
  int carry_sum = 0, carry_min = 0, best = 0;
  for (int k0 = 0; k0 < n; k0 += GM_WAVE) {
    const int k = k0 + lane;
    int sv = 0;
    if (k < n) {
      const uint64_t p = cbase + (uint64_t)g_left + (uint64_t)k;
      const uint32_t gc = (genome[p >> 3] >> ((p & 7) * 4)) & 0xf;
      sv = (gc == (uint32_t)qr[r_left + k]) ? sc.match : sc.mismatch;
    }
    int ps = sv;
    for (int d = 1; d < GM_WAVE; d <<= 1) { const int o = __shfl_up(ps, d); if (lane >= d) ps += o; }
    ps += carry_sum;
    int pm = ps;
    for (int d = 1; d < GM_WAVE; d <<= 1) { const int o = __shfl_up(pm, d); if (lane >= d) pm = min(pm, o); }
    int excl = __shfl_up(pm, 1); if (lane == 0) excl = INT_MAX;
    excl = min(excl, carry_min);
    if (k < n) best = max(best, ps - excl);
    carry_sum = __shfl(ps, GM_WAVE - 1);
    carry_min = min(carry_min, __shfl(pm, GM_WAVE - 1));
  }
  for (int d = 32; d > 0; d >>= 1) best = max(best, __shfl_xor(best, d));
  return best;
}

// hash_genome_window % f1_window_cache_size (ref: common/util.h:224-245, common/hash.h:70-95, f1-wrapper.h:27)
__device__ uint32_t window_hash_slot(const uint8_t* db, int glen, int lane) {
  const int nbuf = (glen + 15) >> 4;
  uint32_t key = 0;
  for (int i0 = 0; i0 < nbuf; i0 += GM_WAVE) {
    const int i = i0 + lane;
    uint32_t buffer = 0;
    if (i < nbuf) for (int j = 0; j < 16 && i * 16 + j < glen; j++) buffer = (buffer << 2) | (db[i * 16 + j] & 3u);
    const int cnt = min(GM_WAVE, nbuf - i0);
    for (int k = 0; k < cnt; k++) {
      const uint32_t bk = (uint32_t)__builtin_amdgcn_readlane((int)buffer, k);
      key += (bk >> 16);
      const uint32_t tmp = ((bk & 0xFFFFu) << 11) ^ key;
      key = (key << 16) ^ tmp;
      key += key >> 11;
    }
  }
  key ^= key << 3; key += key >> 5; key ^= key << 4; key += key >> 17; key ^= key << 25; key += key >> 6;
  return key & (1048576u - 1u);
}

// ---- colour space helpers -------------------------------------------------------------------------
__device__ __forceinline__ int cs_lstocs(int a, int b, bool rna = false) {           // ref: common/util.h:182-205; with is_rna a U counts as T
  if (rna) { a = a == 4 ? 3 : a; b = b == 4 ? 3 : b; }
  return (a > 3 || b > 3) ? 15 : (a ^ b);             // colourmat[a][b] == a ^ b
}
__device__ __forceinline__ int cs_cstols(int first_letter, int colour, bool rna = false) {   // ref: common/util.h:157-180; with is_rna a U goes in as T and a T comes out as U
  if (first_letter == 15 || colour < 0 || colour > 3) return 15;
  if (rna && first_letter == 4) first_letter = 3;
  const int ret = (first_letter % 2 == 0) ? ((4 + first_letter + colour) % 4) : ((4 + first_letter - colour) % 4);
  return (rna && ret == 3) ? 4 : ret;
}

// The window a colour-space hit is scored on (ref: mapping.c:1297-1319): db = colours, db0[c] = lstocs(letter c, primer) for the
// first-colour row (ref: sw-vector.c:116-146).  rc = the hit was turned onto the read's input strand (reverse_hit): the window
// then lives on the reverse-complement contig, whose colour translation is the forward one read backwards, shifted by one
// (complementing both letters keeps their colour), with 'T' + complement(last letter) at its very first position.
__device__ void load_window_cs(const GmIndexDev& ix, int cn, uint32_t goff, int w_len, bool rc, int initbp, uint8_t* db, uint8_t* db0, int lane) {
  const uint64_t cbase = ix.contig_off[cn]; const uint64_t clen = (uint64_t)ix.contig_off[cn + 1] - cbase;
  const bool crna = ix.contig_rna && ix.contig_rna[cn], grna = ix.genome_is_rna != 0;      // the contig's own flag (its reverse complement and colours), the genome's (the SW call)
  const uint64_t cm = gm_cmpl_tab(crna);   // complement_base as nibbles (ref: util.h:125-151)
  for (int c = lane; c < w_len; c += GM_WAVE) {
    uint32_t col, let;
    if (!rc) {
      const uint64_t p = cbase + goff + (uint64_t)c;
      col = (ix.genome_cs[p >> 3] >> ((p & 7) * 4)) & 0xf; let = (ix.genome[p >> 3] >> ((p & 7) * 4)) & 0xf;
    } else {
      const uint64_t q = (uint64_t)goff + (uint64_t)w_len - (uint64_t)c;      // forward colour index; letter index q - 1
      const uint64_t pl = cbase + q - 1;
      let = (ix.genome[pl >> 3] >> ((pl & 7) * 4)) & 0xf; let = (uint32_t)(cm >> (let * 4)) & 0xf;
      if (q == clen) col = (uint32_t)cs_lstocs(3, (int)let, crna);
      else {
        // the colour between the complements of forward letters q and q - 1.  For A / C / G / T that is the forward colour q; a 'U' in the contig complements to 'A'
        // (util.h:125-151), a regular letter, so its colours on the reverse-complement contig are real ones where the forward translation has 15 (fasta.c:586-606)
        const uint64_t pn = cbase + q;
        const uint32_t nxt = (uint32_t)(cm >> (((ix.genome[pn >> 3] >> ((pn & 7) * 4)) & 0xf) * 4)) & 0xf;
        col = (uint32_t)cs_lstocs((int)nxt, (int)let, crna);
      }
    }
    db[c] = (uint8_t)col; db0[c] = (uint8_t)cs_lstocs((int)let, initbp, grna);
  }
}

__device__ __forceinline__ int thr_of(double frac, int absval, int base) { return frac < 0 ? absval : (int)((double)base * frac); }

// ---------------------------------------------------------------------------------------------
// K3: pass 1.  One wave per read-strand walks its windows in (contig, g_off) order.
// ---------------------------------------------------------------------------------------------
// sw_gapless in colour space (ref: common/sw-gapless.c:57-117 with genome_ls != NULL; f1_run's ungapped branch for gmapper-cs -U, mapping.c:1297-1319): the best
// ungapped segment on the diagonal through (g_idx, r_idx) of the contig's colour translation -- of the reverse-complement contig when the hit was turned onto
// the read's input strand (rc) -- with the read's first colour forced against lstocs(letter, primer) when the diagonal starts at read position 0.  Colours and
// letters of the reverse-complement contig come from the forward arrays (see load_window_cs).  sc.mismatch is match + crossover here (gmapper.c:2935).
__device__ int sw_gapless_cs_wave(const GmIndexDev& ix, int cn, bool rc, const uint8_t* qr, int rlen, long long g_idx, int r_idx, int initbp, const GmScoreDev& sc, int lane) {
  const uint64_t cbase = ix.contig_off[cn]; const long long clen = (long long)ix.contig_off[cn + 1] - (long long)cbase;
  const bool crna = ix.contig_rna && ix.contig_rna[cn], grna = ix.genome_is_rna != 0;
  const uint64_t cm = gm_cmpl_tab(crna);   // complement_base as nibbles (ref: util.h:125-151)
  auto nib = [&](const uint32_t* a, uint64_t p) -> int { return (int)((a[p >> 3] >> ((p & 7) * 4)) & 0xf); };
  auto letter_at = [&](long long g) -> int { return rc ? (int)((cm >> (nib(ix.genome, cbase + (uint64_t)(clen - 1 - g)) * 4)) & 0xf) : nib(ix.genome, cbase + (uint64_t)g); };
  auto colour_at = [&](long long g) -> int {
    if (!rc) return nib(ix.genome_cs, cbase + (uint64_t)g);
    if (g == 0) return cs_lstocs(3, letter_at(0), crna);
    return cs_lstocs(letter_at(g - 1), letter_at(g), crna);          // from the complemented letters themselves (a 'U' complements to 'A': see load_window_cs)
  };
  long long g_left; int r_left;
  if (g_idx < r_idx) { g_left = 0; r_left = (int)(r_idx - g_idx); } else { g_left = g_idx - r_idx; r_left = 0; }
  int head = 0;
  if (r_left == 0) {                                   // forcefully match the first colour of the read (:84-94)
    if (g_left < clen) head = (cs_lstocs(letter_at(g_left), initbp, grna) == (int)qr[0]) ? sc.match : 0;
    g_left++; r_left = 1;
  }
  const long long room = clen - g_left;
  const int m = (int)(room < (long long)(rlen - r_left) ? room : (long long)(rlen - r_left));
  int carry_sum = head, carry_min = 0, best = head;
  for (int k0 = 0; k0 < m; k0 += GM_WAVE) {
    const int k = k0 + lane; int sv = 0;
    if (k < m) sv = (colour_at(g_left + k) == (int)qr[r_left + k]) ? sc.match : sc.mismatch;
    int ps = sv;
    for (int d = 1; d < GM_WAVE; d <<= 1) { const int o = __shfl_up(ps, d); if (lane >= d) ps += o; }
    ps += carry_sum;
    int pm = ps;
    for (int d = 1; d < GM_WAVE; d <<= 1) { const int o = __shfl_up(pm, d); if (lane >= d) pm = min(pm, o); }
    int excl = __shfl_up(pm, 1); if (lane == 0) excl = INT_MAX;
    excl = min(excl, carry_min);
    if (k < m) best = max(best, ps - excl);
    carry_sum = __shfl(ps, GM_WAVE - 1);
    carry_min = min(carry_min, __shfl(pm, GM_WAVE - 1));
  }
  for (int d = 32; d > 0; d >>= 1) best = max(best, __shfl_xor(best, d));
  return best;
}

template <bool CS>
__global__ void __launch_bounds__(GM_WAVE) __attribute__((amdgpu_waves_per_eu(8, 8)))      // 64 registers: two of these waves per SIMD fit beside the seed lookup's four (4 x 96 of 512)
k_pass1(GmIndexDev ix, GmScoreDev sc, const uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words,
        int window_len, int overlap_abs, GmHit* __restrict__ hits, const uint16_t* __restrict__ perm,
        const uint32_t* __restrict__ hit_cnt, int hcap, unsigned long long* __restrict__ slots, unsigned long long* __restrict__ stats,
        const int32_t* __restrict__ pair_min, const uint8_t* __restrict__ saved, const uint8_t* __restrict__ initbp, const int early) {
  extern __shared__ __align__(16) uint8_t sm[];
  const int lane = threadIdx.x;
  const int rs = blockIdx.x, rd = rs >> 1, st = rs & 1;
  const int nh = (int)min(hit_cnt[rs], (uint32_t)hcap);
  if (nh == 0) return;
  const int max_w = window_len;
  uint8_t* qr = sm;                                   // read_len
  uint8_t* db = sm + ((read_len + 15) & ~15);         // max_w
  uint8_t* db0 = db + ((max_w + 15) & ~15);           // colour space: first-colour row (max_w)
  int16_t* carry = (int16_t*)(db0 + (CS ? ((max_w + 15) & ~15) : 0));
  // colour space scores every window against the read as it was sequenced (strand 0): the hit is reversed instead (ref: mapping.c:1302-1303)
  load_read(reads + (size_t)rd * read_words, read_len, CS ? false : (st != 0), qr, lane, !CS && ix.read_rna && ix.read_rna[rd]);
  const int ib = CS ? (int)initbp[rd] : 0;
  __syncthreads();
  GmHit* H = hits + (size_t)rs * hcap;
  const uint16_t* P = perm + (size_t)rs * hcap;
  unsigned long long* SL = slots + (size_t)rs * hcap; // (cache slot, score) of every computed window (hash_filter_calls); sc1 accesses: written by lane 0, read by all
  int last_good_cn = -1; uint32_t last_good_goff = 0;
  int n_comp = 0;
  unsigned long long calls = 0, cells = 0, bypass = 0;
  for (int t = 0; t < nh; t++) {
    const int hi = P[t];
    GmHit* h = &H[hi];
    const int matches = h->matches; const int cn = h->cn; const uint32_t goff = h->g_off; const int w_len = h->w_len;
    if (pair_min && pair_min[(size_t)rs * hcap + t] < 0) continue;                      // only_paired, ref: mapping.c:1271-1273
    if (matches < sc.min_matches) continue;                                             // ref: mapping.c:1275
    if (saved && saved[(size_t)rs * hcap + hi]) { last_good_cn = cn; last_good_goff = goff; continue; }   // ref :1279-1284
    if (last_good_cn >= 0 && cn == last_good_cn &&
        (long long)goff + (long long)(uint32_t)overlap_abs <= (long long)(uint32_t)(last_good_goff + (uint32_t)window_len)) {  // ref :1287-1293
      if (lane == 0) { h->score_vector = 0; h->pct_score_vector = 0; }
      continue;
    }
    // ref :1295 -- a window keeps a positive score from an earlier pass (paired mode runs pass 1 twice); unpaired: always <= 0 here
    if (h->score_vector > 0) continue;
    const uint64_t g0 = (uint64_t)ix.contig_off[cn] + goff;
    if (CS) load_window_cs(ix, cn, goff, w_len, st != ix.cs_flip, ib, db, db0, lane);      // the hit is turned onto the read's input strand (label cs_flip)
    else load_window(ix.genome, g0, w_len, false, db, lane);
    __syncthreads();
    int score = -1; bool computed = false, was_cut = false;
    uint32_t slot = 0;
    if (sc.hash_filter_calls) {                                                          // f1_run look-up, ref: f1-wrapper.h:103-114
      slot = window_hash_slot(db, w_len, lane);
      // first computed window of this read-strand with the same slot wins (same tag)
      int found = -1;
      for (int c0 = 0; c0 < n_comp; c0 += GM_WAVE) {
        const int c = c0 + lane;
        const bool hit = (c < n_comp) && ((uint32_t)__hip_atomic_load(&SL[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == slot);
        const unsigned long long bal = __ballot(hit);
        if (bal) { found = c0 + __builtin_ctzll(bal); break; }
      }
      if (found >= 0) {
        const unsigned long long e = __hip_atomic_load(&SL[found], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        score = (int)((e >> 32) & 0x7FFFu); bypass++;
        if ((e >> 47) & 1u) {
          // the entry holds a lower bound of a score below ITS window's threshold (early stop); a window with a lower threshold (shorter than the
          // read) needs the value itself: score the entry's window in full and put it right
          const GmHit* ha = &H[(int)(e >> 48)];
          const int a_len = ha->w_len;
          const int a_max = (read_len < a_len ? read_len : a_len) * sc.match, b_max = (read_len < w_len ? read_len : w_len) * sc.match;
          if (thr_of(sc.vect_thr_frac, sc.vect_abs, b_max) < thr_of(sc.vect_thr_frac, sc.vect_abs, a_max)) {
            __syncthreads();
            if (CS) load_window_cs(ix, ha->cn, ha->g_off, a_len, st != ix.cs_flip, ib, db, db0, lane);
            else load_window(ix.genome, (uint64_t)ix.contig_off[ha->cn] + ha->g_off, a_len, false, db, lane);
            __syncthreads();
            score = sw_vector_wave_t<CS>(db, db0, a_len, qr, read_len, sc, carry, lane);
            if (lane == 0) __hip_atomic_store(&SL[found], (e & 0xFFFF0000FFFFFFFFull) | ((unsigned long long)(uint32_t)score << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
          }
        }
      }
    }
    if (score < 0) {
      if (!CS && sc.gapless) {                                                           // -U: f1_run's ungapped branch, ref: f1-wrapper.h:122-125, mapping.c:1321-1328
        score = sw_gapless_wave(ix.genome, (uint64_t)ix.contig_off[cn], (long long)ix.contig_off[cn + 1] - ix.contig_off[cn], qr, read_len,
                                (long long)goff + h->ax, h->ay, sc, lane);
        calls++; cells += (unsigned long long)read_len;
      } else if (CS && sc.gapless) {                                                     // the same in colour space: the hit is first turned onto the read's input strand (mapping.c:1302-1303)
        const bool rcw = st != ix.cs_flip;
        const long long clen = (long long)ix.contig_off[cn + 1] - ix.contig_off[cn];
        long long go = goff; long long ax = h->ax, ay = h->ay;
        if (rcw) {                                                                       // reverse_hit + anchor_reverse, ref: mapping.c:254-263, anchors.h:30-34
          go = clen - go - w_len;
          ax = -ax + (w_len - 1) - (h->alen - 1) - (h->awidth - 1);
          ay = -ay + (read_len - 1) - (h->alen - 1) + (h->awidth - 1);
        }
        score = sw_gapless_cs_wave(ix, cn, rcw, qr, read_len, go + ax, (int)ay, ib, sc, lane);
        calls++; cells += (unsigned long long)read_len;
      } else {
        const int score_max_w = (read_len < w_len ? read_len : w_len) * sc.match;
        score = sw_vector_wave_t<CS>(db, db0, w_len, qr, read_len, sc, carry, lane, early ? thr_of(sc.vect_thr_frac, sc.vect_abs, score_max_w) : 0, &was_cut); computed = true;
        calls++; cells += (unsigned long long)w_len * read_len;
      }
      if (sc.hash_filter_calls) {
        // (cache slot | score, 15 bits | early stop | hit index)
        if (lane == 0) __hip_atomic_store(&SL[n_comp], (unsigned long long)slot | ((unsigned long long)(uint32_t)score << 32) | ((unsigned long long)(was_cut ? 1u : 0u) << 47) | ((unsigned long long)(uint32_t)hi << 48),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        n_comp++; __syncthreads();
      }
    }
    const int score_max = (read_len < w_len ? read_len : w_len) * sc.match;
    // flags bit 0: the score was computed for this very window (not taken from the f1 cache): pass 2 need not re-score it
    if (lane == 0) { h->score_vector = score; h->pct_score_vector = (1000 * 100 * score) / score_max; h->flags = (uint16_t)((h->flags & ~1u) | (computed ? 1u : 0u)); }
    if (score >= thr_of(sc.vect_thr_frac, sc.vect_abs, score_max)) { last_good_cn = cn; last_good_goff = goff; }   // ref :1332-1335
    __syncthreads();
  }
  if (lane == 0) {
    GS_ADD(stats, GS_VEC_CALLS, calls); GS_ADD(stats, GS_VEC_CELLS, cells); GS_ADD(stats, GS_VEC_BYPASSED, bypass);
  }
}

// ---------------------------------------------------------------------------------------------
// K4a: top-K by pass1 key with the reference's ext-heap (array order matters downstream).
// One thread per read; sel[rd][k] = (st << 16) | hit index, in heap array order.
// ---------------------------------------------------------------------------------------------
#define SEL_MAX 64
__global__ void __launch_bounds__(64)
k_select(GmScoreDev sc, int n_reads, int read_len, const GmHit* __restrict__ hits, const uint16_t* __restrict__ perm,
         const uint32_t* __restrict__ hit_cnt, int hcap, int32_t* __restrict__ sel, uint32_t* __restrict__ sel_cnt,
         const uint8_t* __restrict__ saved) {
  const int rd = blockIdx.x * blockDim.x + threadIdx.x;
  if (rd >= n_reads) return;
  int key[SEL_MAX]; int id[SEL_MAX];
  int load = 0;
  const int K = min(sc.num_tmp_outputs, SEL_MAX);
  const bool absthr = sc.vect_thr_frac < 0;
  for (int st = 0; st < 2; st++) {
    const int rs = rd * 2 + st;
    const int nh = (int)min(hit_cnt[rs], (uint32_t)hcap);
    const GmHit* H = hits + (size_t)rs * hcap;
    const uint16_t* P = perm + (size_t)rs * hcap;
    for (int t = 0; t < nh; t++) {
      const GmHit& h = H[P[t]];
      if (saved && saved[(size_t)rs * hcap + P[t]]) continue;                             // ref: mapping.c:1388
      const int score_max = (read_len < (int)h.w_len ? read_len : (int)h.w_len) * sc.match;
      const int k = absthr ? h.score_vector : h.pct_score_vector;
      if (h.score_vector >= thr_of(sc.vect_thr_frac, sc.vect_abs, score_max) && (load < K || k > key[0])) {   // ref: mapping.c:1391-1396
        const int me = (st << 16) | (int)P[t];
        if (load < K) {                       // extheap insert + percolate_up (strict <)
          key[load] = k; id[load] = me; load++;
          int node = load, parent = node / 2;
          while (node > 1 && key[node - 1] < key[parent - 1]) {
            int tk = key[parent - 1]; key[parent - 1] = key[node - 1]; key[node - 1] = tk;
            int ti = id[parent - 1]; id[parent - 1] = id[node - 1]; id[node - 1] = ti;
            node = parent; parent = node / 2;
          }
        } else {                              // replace_min + percolate_down
          key[0] = k; id[0] = me;
          int node = 1;
          for (;;) {
            int left = node * 2, right = left + 1, mn = node;
            if (left <= load && key[left - 1] < key[node - 1]) mn = left;
            if (right <= load && key[right - 1] < key[mn - 1]) mn = right;
            if (mn == node) break;
            int tk = key[mn - 1]; key[mn - 1] = key[node - 1]; key[node - 1] = tk;
            int ti = id[mn - 1]; id[mn - 1] = id[node - 1]; id[node - 1] = ti;
            node = mn;
          }
        }
      }
    }
  }
  for (int k = 0; k < load; k++) sel[(size_t)rd * SEL_MAX + k] = id[k];
  sel_cnt[rd] = (uint32_t)load;
}

// ---------------------------------------------------------------------------------------------
// K4b: pass 2.  One wave per selected hit: reverse_hit when needed, re-score with the vector
// filter, then the banded 3-state full SW with the reference's tie rules, and the backtrace.
// Lane l owns read row 64s + l; column t - l at step t.  Cells outside the band are -INF, as the
// reference's init_cell(.., 0) leaves them (sw-full-ls.c:66-80,229-231,378-385); the virtual row
// above the matrix is init_cell(.., 1): nw 0, n -b_open, w -a_open, back 0 (:194-196).
// ---------------------------------------------------------------------------------------------
#define FS_NEG (-(INT_MAX / 2))

__device__ __forceinline__ void band_range(long long ax, long long ay, int alen, int awidth, int x_len, int y, int* x_min, int* x_max) {
  // anchor_get_x_range, ref: common/anchors.c:64-95
  int mn, mx;
  if (y < ay) mn = 0;
  else if (y <= ay + (alen - 1)) mn = (int)(ax + (y - ay));
  else mn = (int)(ax + alen);
  if (mn < 0) mn = 0;
  if (mn >= x_len) mn = x_len - 1;
  if (y < ay - (awidth - 1)) mx = (int)(ax + (awidth - 1) - 1);
  else if (y <= ay - (awidth - 1) + (alen - 1)) mx = (int)(ax + (awidth - 1) + (y - (ay - (awidth - 1))));
  else mx = x_len - 1;
  if (mx < 0) mx = 0;
  if (mx >= x_len) mx = x_len - 1;
  *x_min = mn; *x_max = mx;
}

__device__ __forceinline__ int shr1_i(int v, int lane0) { return __builtin_amdgcn_update_dpp(lane0, v, 0x138, 0xf, 0xf, false); }

struct FullOut { int score, max_i, max_j, e_nw, e_n, e_w; };

// back byte: bits 0-1 nw source (0 NW_NW, 1 NW_N, 2 NW_W), bit 2 n source (0 N_NW, 1 N_N), bit 3 w source (0 W_NW, 1 W_W), bit 7 = cell computed;
// local mode: bits 4 / 5 / 6 = the nw / n / w state was floored at 0, i.e. carries the reference's null back pointer.
// LOCAL (Gflag off, ref: sw-full-ls.c:66-80,293-374): every cell outside the band that a band cell can read holds (0, -b_open, -a_open);
// a state <= 0 becomes 0 with a null back pointer; the result is the first cell in row-major order with the largest score.
template <bool LOCAL>
__device__ FullOut full_sw_wave(const uint8_t* db, int glen, const uint8_t* qr, int rlen, const GmScoreDev& sc, bool revcmpl,
                                long long rx, long long ry, int rl, int rw, uint8_t* back, int* carry, int lane) {
  const int a_go = sc.a_go, a_ge = sc.a_ge, b_go = sc.b_go, b_ge = sc.b_ge;
  const int o_nw = LOCAL ? 0 : FS_NEG, o_n = LOCAL ? -b_go : FS_NEG, o_w = LOCAL ? -a_go : FS_NEG;   // a cell outside the band
  FullOut out; out.score = 0; out.max_i = 0; out.max_j = 0; out.e_nw = out.e_n = out.e_w = 0;
  const int n_stripes = (rlen + 63) >> 6;
  int* cNW = carry; int* cN = carry + glen; int* cW = carry + 2 * glen;   // last row of the previous stripe, per column
  int cw_lo = 1, cw_hi = 0;                                    // columns of the carry rows the previous stripe wrote (the others read as out-of-band cells)
  for (int s = 0; s < n_stripes; s++) {
    const int r = s * 64 + lane;
    const bool row_ok = r < rlen;
    const int q = row_ok ? qr[r] : 0x7F;
    int x_min = 0, x_max = -1;
    if (row_ok) band_range(rx, ry, rl, rw, glen, r, &x_min, &x_max);
    // Only the steps at which some lane stands inside the band: the band is a strip along the diagonal (the anchor box widened by anchor_width), so
    // column t - lane is inside it for t in [min(x_min + lane), max(x_max + lane)] -- ~150 of the glen + rows - 1 ~ 200 steps of a 100 bp read.  Before
    // and after, every lane holds the constants of an out-of-band cell: nothing to compute.
    int t_lo = INT_MAX, t_hi = -1;
    if (row_ok && x_max >= x_min) { t_lo = x_min + lane; t_hi = x_max + lane; }
    for (int dd = 32; dd > 0; dd >>= 1) { t_lo = min(t_lo, __shfl_xor(t_lo, dd)); t_hi = max(t_hi, __shfl_xor(t_hi, dd)); }
    // own previous cell (r, c-1) and the two cells of row r-1 needed next: (r-1, c) arrives by shift
    int pw_nw = o_nw, pw_n = o_n, pw_w = o_w;                  // cell_w  = (r, c-1); column -1 is out of band
    int d_nw = o_nw, d_n = o_n, d_w = o_w;                     // cell_nw = (r-1, c-1)
    int cur_nw = o_nw, cur_n = o_n, cur_w = o_w;               // this lane's latest cell, shifted to lane+1 next step
    if (lane == 0) {
      if (s == 0) { d_nw = 0; d_n = -b_go; d_w = -a_go; }      // virtual row -1 (every column of it, -1 included)
      else if (t_hi >= 0 && t_lo >= 1 && t_lo - 1 >= cw_lo && t_lo - 1 <= cw_hi) { d_nw = cNW[t_lo - 1]; d_n = cN[t_lo - 1]; d_w = cW[t_lo - 1]; }   // (r-1, t_lo-1) of the previous stripe's last row
    }
    const bool more = (s + 1 < n_stripes);
    const bool last_row_lane = row_ok && (r == rlen - 1);
    for (int t = t_lo; t <= t_hi; t++) {
      const int c = t - lane;
      // lane 0's upper neighbour at column t: virtual row (stripe 0) or carry from the previous stripe
      int in_nw, in_n, in_w;
      if (s == 0) { in_nw = 0; in_n = -b_go; in_w = -a_go; }
      else { in_nw = o_nw; in_n = o_n; in_w = o_w; if (t >= cw_lo && t <= cw_hi) { in_nw = cNW[t]; in_n = cN[t]; in_w = cW[t]; } }
      const int u_nw = shr1_i(cur_nw, in_nw), u_n = shr1_i(cur_n, in_n), u_w = shr1_i(cur_w, in_w);   // cell_n = (r-1, c)
      const bool inband = row_ok && c >= x_min && c <= x_max;
      int n_nw = o_nw, n_n = o_n, n_w = o_w;
      if (inband) {
        const int ms = (db[c] == q) ? sc.match : sc.mismatch;
        int tmp, b0, b1, b2, nul = 0;
        if (!revcmpl) {                                            // ref: sw-full-ls.c:264-278
          tmp = d_nw + ms; b0 = 0;
          if (d_n + ms > tmp) { tmp = d_n + ms; b0 = 1; }
          if (d_w + ms > tmp) { tmp = d_w + ms; b0 = 2; }
        } else {                                                   // :279-292
          tmp = d_w + ms; b0 = 2;
          if (d_n + ms > tmp) { tmp = d_n + ms; b0 = 1; }
          if (d_nw + ms > tmp) { tmp = d_nw + ms; b0 = 0; }
        }
        if (LOCAL && tmp <= 0) { tmp = 0; nul |= 0x10; }
        n_nw = tmp;
        if (!revcmpl) {                                            // north :303-320
          tmp = u_nw - b_go - b_ge; b1 = 0;
          if (u_n - b_ge > tmp) { tmp = u_n - b_ge; b1 = 1; }
        } else {
          tmp = u_n - b_ge; b1 = 1;
          if (u_nw - b_go - b_ge > tmp) { tmp = u_nw - b_go - b_ge; b1 = 0; }
        }
        if (LOCAL && tmp <= 0) { tmp = 0; nul |= 0x20; }
        n_n = tmp;
        if (!revcmpl) {                                            // west :330-347
          tmp = pw_nw - a_go - a_ge; b2 = 0;
          if (pw_w - a_ge > tmp) { tmp = pw_w - a_ge; b2 = 1; }
        } else {
          tmp = pw_w - a_ge; b2 = 1;
          if (pw_nw - a_go - a_ge > tmp) { tmp = pw_nw - a_go - a_ge; b2 = 0; }
        }
        if (LOCAL && tmp <= 0) { tmp = 0; nul |= 0x40; }
        n_w = tmp;
        back[(size_t)r * glen + c] = (uint8_t)(0x80 | nul | b0 | (b1 << 2) | (b2 << 3));
        if (LOCAL || last_row_lane) {                              // :359-368 leftmost strict maximum: of the last read row, or (local) of this lane's rows
          int m = max(n_n, n_nw); m = max(m, n_w);
          if (m > out.score) { out.score = m; out.max_i = r; out.max_j = c; out.e_nw = n_nw; out.e_n = n_n; out.e_w = n_w; }
        }
      }
      if (more && lane == 63 && c >= 0 && c < glen) { cNW[c] = n_nw; cN[c] = n_n; cW[c] = n_w; }
      // advance: the cell just computed becomes cell_w; the shifted-in cell becomes next step's cell_nw
      d_nw = u_nw; d_n = u_n; d_w = u_w;
      pw_nw = n_nw; pw_n = n_n; pw_w = n_w;
      cur_nw = n_nw; cur_n = n_n; cur_w = n_w;
      (void)pw_n;
    }
    cw_lo = 1; cw_hi = 0;
    if (more && t_hi >= 0) { cw_lo = max(0, t_lo - 63); cw_hi = min(glen - 1, t_hi - 63); }
    if (more) __syncthreads();
  }
  // broadcast the result: from the last row's lane, or (local) from the lane whose row comes first among those with the largest score
  int src = (rlen - 1) & 63;
  if (LOCAL) {
    int best = out.score;
    for (int d = 32; d > 0; d >>= 1) best = max(best, __shfl_xor(best, d));
    int row = (out.score == best) ? out.max_i : INT_MAX;
    for (int d = 32; d > 0; d >>= 1) row = min(row, __shfl_xor(row, d));
    src = (best > 0) ? (row & 63) : 0;
  }
  out.score = __shfl(out.score, src); out.max_i = __shfl(out.max_i, src); out.max_j = __shfl(out.max_j, src);
  out.e_nw = __shfl(out.e_nw, src); out.e_n = __shfl(out.e_n, src); out.e_w = __shfl(out.e_w, src);
  return out;
}

// The band sw_full_ls falls back to without anchors (ref: sw-full-ls.c:179-192): anchor_join of the corner anchors (0, y0) and
// (glen - 1, rlen - 1 - y0), y0 = (rlen * match - thresh) / match, both of length and width 1 (anchors.c:9-52).
__device__ __forceinline__ void threshold_band(int glen, int rlen, int match, int thresh, long long* rx, long long* ry, int* rl, int* rw) {
  const long long y0 = ((long long)rlen * match - thresh) / match;
  const long long bx[2] = {0, glen - 1}, by[2] = {y0, rlen - 1 - y0};
  long long b_nw = bx[0] + by[0], b_sw = bx[0] - by[0], b_ne = b_sw, b_se = b_nw;
  b_nw = bx[1] + by[1] < b_nw ? bx[1] + by[1] : b_nw; b_sw = bx[1] - by[1] < b_sw ? bx[1] - by[1] : b_sw;
  b_ne = bx[1] - by[1] > b_ne ? bx[1] - by[1] : b_ne; b_se = bx[1] + by[1] > b_se ? bx[1] + by[1] : b_se;
  if ((b_nw + b_sw) % 2 != 0) b_nw--;
  *rx = (b_nw + b_sw) / 2; *ry = b_nw - *rx;
  if ((b_ne - b_sw) % 2 != 0) b_ne++;
  *rw = (int)((b_ne - b_sw) / 2 + 1);
  if ((b_se - b_nw) % 2 != 0) b_se++;
  *rl = (int)((b_se - b_nw) / 2 + 1);
}

template <bool BACK_LDS, bool LOCAL>
__global__ void __launch_bounds__(GM_WAVE)
k_pass2(GmIndexDev ix, GmScoreDev sc, const uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words,
        GmHit* __restrict__ hits, const uint16_t* __restrict__ perm, int hcap,
        const int32_t* __restrict__ sel, const int32_t* __restrict__ sel_sidx, int input_strand, int write_back,
        const uint32_t* __restrict__ sel_cnt,
        const uint32_t* __restrict__ work, const uint32_t* __restrict__ n_work_p,
        GmFullRes* __restrict__ res, uint8_t* __restrict__ ops, int ops_stride,
        uint8_t* __restrict__ back_pool, size_t back_stride, int max_w, unsigned long long* __restrict__ stats, int ablate) {
  extern __shared__ __align__(16) uint8_t sm[];
  const int lane = threadIdx.x;
  uint8_t* qr = sm;
  uint8_t* db = sm + ((read_len + 15) & ~15);
  int* carry = (int*)(db + ((max_w + 15) & ~15));
  // back pointers: one byte per cell; in LDS when read_len x window fits (short reads), else in a per-wave global scratch
  uint8_t* back = BACK_LDS ? (uint8_t*)(carry + 3 * max_w) : (back_pool + (size_t)blockIdx.x * back_stride);
  const uint32_t n_work = *n_work_p;
  unsigned long long vcalls = 0, vcells = 0, fcalls = 0;
  int cur_rd = -1;
  for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
    const uint32_t wk = work[wi];
    const int rd = (int)(wk >> 6), k = (int)(wk & 63);
    const int id = sel[(size_t)rd * SEL_MAX + k];
    int st = id >> 16; const int hi = id & 0xFFFF;
    const GmHit h = hits[((size_t)rd * 2 + st) * hcap + hi];
    // the read in its input orientation == read[input_strand]; a read_reverse'd mate (ref: gmapper.c:174-185) is stored reverse-complemented
    if (rd != cur_rd) { __syncthreads(); load_read(reads + (size_t)rd * read_words, read_len, input_strand != 0, qr, lane, ix.read_rna && ix.read_rna[rd]); cur_rd = rd; }
    const int cn = h.cn, w_len = h.w_len;
    const long long clen = (long long)ix.contig_off[cn + 1] - ix.contig_off[cn];
    long long g_off = h.g_off; long long ax = h.ax, ay = h.ay; int gen_st = 0;
    const size_t slot = ((size_t)rd * 2 + st) * hcap + hi;
    if (st != input_strand) {                                   // reverse_hit, ref: mapping.c:254-263,337-339; anchor_reverse anchors.h:30-34
      g_off = clen - g_off - w_len;
      ax = -ax + (w_len - 1) - (h.alen - 1) - (h.awidth - 1);
      ay = -ay + (read_len - 1) - (h.alen - 1) + (h.awidth - 1);
      gen_st = 1; st = input_strand;
    }
    // the window on the gen_st strand == the + strand window of the original hit, reverse-complemented
    const uint64_t g0 = (uint64_t)ix.contig_off[cn] + h.g_off;
    __syncthreads();
    load_window(ix.genome, g0, w_len, gen_st != 0, db, lane, ix.contig_rna && ix.contig_rna[cn]);
    __syncthreads();
    const int score_max = (read_len < w_len ? read_len : w_len) * sc.match;
    const int thresh = thr_of(sc.full_thr_frac, sc.full_abs, score_max);
    // ref: mapping.c:386-388 re-scores because the pass-1 value may come from the f1 cache (another window with the same hash slot).
    // When pass 1 computed it for this very window the re-score is the same number: local SW with direction-free gap costs is
    // invariant under reversing + complementing both sequences (N and IUPAC codes map one-to-one), which is all reverse_hit does.
    int sv;
    if ((h.flags & 1u) && !GM_ABL(4)) sv = h.score_vector;
    else { sv = sw_vector_wave(db, w_len, qr, read_len, sc, (int16_t*)carry, lane); vcalls++; vcells += (unsigned long long)w_len * read_len; }
    GmFullRes R;
    R.read_idx = rd; R.st = (int16_t)ix.cs_flip; R.gen_st = (int16_t)gen_st; R.cn = (uint32_t)cn; R.g_off = (uint32_t)g_off; R.w_len = w_len;
    R.score_vector = sv; R.score_max = score_max; R.matches = h.matches; R.score_window_gen = h.score_window_gen;
    R.score = 0; R.read_start = 0; R.rmapped = 0; R.genome_start = 0; R.gmapped = 0;
    R.n_match = R.n_mismatch = R.n_ins = R.n_del = 0; R.n_ops = 0; R.ops_off = (uint32_t)(wi * (uint32_t)ops_stride);
    R.sort_idx = sel_sidx ? sel_sidx[(size_t)rd * SEL_MAX + k] : 0; R.hit_slot = (uint32_t)slot;
    if (write_back && lane == 0) hits[slot].score_vector = sv;          // hit_run_full_sw keeps the re-scored value in the hit (ref: mapping.c:386-388)
    if (sv >= thresh && !GM_ABL(1)) {
      fcalls++;
      // rectangle = anchor_join(1 anchor) + anchor_widen(anchor_width), ref: sw-full-ls.c:176-178, anchors.c:9-61
      long long nw = ax + ay, sw = ax - ay, ne = sw + 2 * (h.awidth - 1), se = nw + 2 * (h.alen - 1);
      if ((nw + sw) % 2 != 0) nw--;
      long long rx = (nw + sw) / 2, ry = nw - rx;
      if ((ne - sw) % 2 != 0) ne++;
      int rw = (int)((ne - sw) / 2 + 1);
      if ((se - nw) % 2 != 0) se++;
      int rl = (int)((se - nw) / 2 + 1);
      rx -= sc.anchor_width / 2; ry += sc.anchor_width / 2; rw += sc.anchor_width;
      __syncthreads();
      FullOut fo = full_sw_wave<LOCAL>(db, w_len, qr, read_len, sc, (gen_st != 0) && sc.tiebreak_rev, rx, ry, rl, rw, back, carry, lane);
      __syncthreads();
      if (LOCAL && fo.score != sv) {
        // the filter's best local alignment leaves the anchor band: once more over the band the threshold allows (ref: sw-full-ls.c:395-398)
        threshold_band(w_len, read_len, sc.match, thresh, &rx, &ry, &rl, &rw);
        fo = full_sw_wave<LOCAL>(db, w_len, qr, read_len, sc, (gen_st != 0) && sc.tiebreak_rev, rx, ry, rl, rw, back, carry, lane);
        __syncthreads();
      }
      R.score = fo.score;
      if (fo.score > 0) {
        // do_backtrace, ref: sw-full-ls.c:413-516 -- lane 0 walks; ops are emitted reversed then flipped
        if (lane == 0 && !GM_ABL(2)) {
          int i = fo.max_i, j = fo.max_j;
          // from-state: 0 nw, 1 n, 2 w  (ref :420-427: nw, then w if strictly greater, then n if strictly greater)
          int state = 0, fs = fo.e_nw;
          if (fo.e_w > fs) { state = 2; fs = fo.e_w; }
          if (fo.e_n > fs) state = 1;
          uint8_t* o = ops + (size_t)R.ops_off;
          int no = 0, rstart = 0, gstart = 0, nm = 0, nmm = 0, nin = 0, ndel = 0;
          while (i >= 0 && j >= 0) {
            const uint8_t bb = BACK_LDS ? back[(size_t)i * w_len + j]
                                        : __hip_atomic_load(&back[(size_t)i * w_len + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(bb & 0x80)) break;                // out-of-band cell: back == 0 in the reference
            if (LOCAL) {                            // a floored state has a null back pointer; a cell outside the band was never computed (stale byte)
              if ((bb >> (4 + state)) & 1) break;
              int bx_min, bx_max; band_range(rx, ry, rl, rw, w_len, i, &bx_min, &bx_max);
              if (j < bx_min || j > bx_max) break;
            }
            int nstate;
            if (state == 1) {                       // FROM_NORTH_*: BACK_DELETION (gap in the genome)
              if (no < ops_stride) o[no] = 'D'; no++; ndel++; rstart = i; i--;
              nstate = ((bb >> 2) & 1) ? 1 : 0;
            } else if (state == 2) {                // FROM_WEST_*: BACK_INSERTION (gap in the read)
              if (no < ops_stride) o[no] = 'I'; no++; nin++; gstart = j; j--;
              nstate = ((bb >> 3) & 1) ? 2 : 0;
            } else {                                // FROM_NORTHWEST_*
              if (no < ops_stride) o[no] = 'M'; no++;
              if (db[j] == qr[i]) nm++; else nmm++;
              rstart = i; gstart = j; i--; j--;
              nstate = (bb & 3);                    // 0 nw, 1 n, 2 w
            }
            state = nstate;
          }
          const int nov = min(no, ops_stride);
          for (int a = 0, b = nov - 1; a < b; a++, b--) { uint8_t tt = o[a]; o[a] = o[b]; o[b] = tt; }
          R.n_ops = no; R.read_start = rstart; R.genome_start = gstart + (int)g_off;
          R.gmapped = fo.max_j - gstart + 1; R.rmapped = fo.max_i - rstart + 1;
          R.n_match = nm; R.n_mismatch = nmm; R.n_ins = nin; R.n_del = ndel;
        }
      }
    }
    if (lane == 0) res[wi] = R;
  }
  if (lane == 0) { GS_ADD(stats, GS_FULL_CALLS, fcalls); GS_ADD(stats, GS_VEC_CALLS, vcalls); GS_ADD(stats, GS_VEC_CELLS, vcells); }
}

// ---------------------------------------------------------------------------------------------
// K4b, four windows per wave (round 3).  The band of the reference's full SW is a strip of ~10-20 columns along the anchor's diagonal (plus two small
// corner blocks), so with one read row per lane of a 64-row stripe only a dozen lanes stand inside the band at any step: k_pass2 issues its vector
// instructions with 48 % of the lanes enabled and a sixth of them computing.  Here a wave takes FOUR selected windows: each group of 16 lanes owns one
// window, lane l of the group the read row 16 s + l of stripe s, the row above arrives by a DPP shift inside the group (row_shr:1; lane 0 of a group gets
// the virtual row / the carry of the previous stripe), and a stripe needs band width + 30 steps instead of band width + 126.  Same cell code, same tie
// rules, same back bytes as full_sw_wave (ref: sw-full-ls.c:154-403); the four group leaders walk their tracebacks side by side.
// Everything that is wave-uniform in full_sw_wave and differs between windows -- band box, window length, strand flag, "this group has work" -- is a
// per-lane value here that is uniform inside a group.
// ---------------------------------------------------------------------------------------------
// Groups of G lanes (G = 16: four windows a wave, G = 8: eight).  row_shr:1 moves inside rows of 16 lanes, so with G = 8 the first lane of a row's second group is put right afterwards.
template <int G> __device__ __forceinline__ int gn_shr1(int v, int first, int l) {
  const int r = __builtin_amdgcn_update_dpp(first, v, 0x111, 0xf, 0xf, false);
  return (G == 16) ? r : (l == 0 ? first : r);
}
// The carry rows (last row of a stripe, per column and state) as int, or as int16_t where the launch has checked that no score of a real path lies further than 16000 from 0: a
// value of the "minus infinity" family (FS_NEG plus a few penalties) is stored as -32768 and goes on from there -- -32768 plus anything a path can collect stays below every
// score of a real path, so such a state never wins against a real one, a state on the traced path never has one as its source, and the alignment is the same
// (gm_launch_pass2 / gm_launch_pass2_cs decide; ref for the values: sw-full-ls.c:66-80,194-196, sw-full-cs.c:201-215,312-322).
__device__ __forceinline__ int cs_carry_ld(int v) { return v; }
__device__ __forceinline__ int cs_carry_ld(int16_t v) { return (int)v; }
__device__ __forceinline__ void cs_carry_st(int& d, int v) { d = v; }
__device__ __forceinline__ void cs_carry_st(int16_t& d, int v) { d = (int16_t)max(v, -32768); }

// Round 4: G lanes a window (8 where the carry rows fit int16_t: a stripe of 8 rows takes band width + 14 steps against band width + 30 for 16 rows -- the chain north, west,
// north, ... through a band is two steps a row whatever the lane count), the strand kind REV a template constant (a pass holds windows of one kind, see k_pass2_g4), and the cell
// worked out by every lane with the band deciding afterwards what is kept (no branch around it).
template <int G, typename CT, bool LOCAL, bool REV>
__device__ FullOut full_sw_g4(const uint8_t* db, int glen, const uint8_t* qr, int rlen, const GmScoreDev& sc, bool act,
                              long long rx, long long ry, int rl, int rw, uint8_t* back, CT* carry, int lane) {
  constexpr bool revcmpl = REV;
  const int a_go = sc.a_go, a_ge = sc.a_ge, b_go = sc.b_go, b_ge = sc.b_ge;
  const int o_nw = LOCAL ? 0 : FS_NEG, o_n = LOCAL ? -b_go : FS_NEG, o_w = LOCAL ? -a_go : FS_NEG;   // a cell outside the band
  FullOut out; out.score = 0; out.max_i = 0; out.max_j = 0; out.e_nw = out.e_n = out.e_w = 0;
  const int l = lane & (G - 1);
  const int n_stripes = (rlen + G - 1) / G;
  CT* cNW = carry; CT* cN = carry + glen; CT* cW = carry + 2 * glen;   // this group's carry rows: last row of the previous stripe, per column
  int cw_lo = 1, cw_hi = 0;
  for (int s = 0; s < n_stripes; s++) {
    const int r = s * G + l;
    const bool row_ok = act && r < rlen;
    const int q = row_ok ? qr[r] : 0x7F;
    int x_min = 0, x_max = -1;
    if (row_ok) band_range(rx, ry, rl, rw, glen, r, &x_min, &x_max);
    int t_lo = INT_MAX, t_hi = -1;
    if (row_ok && x_max >= x_min) { t_lo = x_min + l; t_hi = x_max + l; }
    for (int dd = G / 2; dd > 0; dd >>= 1) { t_lo = min(t_lo, __shfl_xor(t_lo, dd)); t_hi = max(t_hi, __shfl_xor(t_hi, dd)); }     // over the group
    int nst = t_hi >= 0 ? t_hi - t_lo + 1 : 0, nmax = nst;
    for (int dd = 32; dd >= G; dd >>= 1) nmax = max(nmax, __shfl_xor(nmax, dd));                                                // over the groups
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    int pw_nw = o_nw, pw_w = o_w;
    int d_nw = o_nw, d_n = o_n, d_w = o_w;
    int cur_nw = o_nw, cur_n = o_n, cur_w = o_w;
    if (l == 0) {
      if (s == 0) { d_nw = 0; d_n = -b_go; d_w = -a_go; }
      else if (t_hi >= 0 && t_lo >= 1 && t_lo - 1 >= cw_lo && t_lo - 1 <= cw_hi) { d_nw = cs_carry_ld(cNW[t_lo - 1]); d_n = cs_carry_ld(cN[t_lo - 1]); d_w = cs_carry_ld(cW[t_lo - 1]); }
    }
    const bool more = (s + 1 < n_stripes);
    const bool last_row_lane = row_ok && (r == rlen - 1);
    for (int i = 0; i < nmax; i++) {
      const bool on = i < nst;                                   // this group still has steps in this stripe
      const int t = t_lo + i;
      const int c = t - l;
      int in_nw, in_n, in_w;
      if (s == 0) { in_nw = 0; in_n = -b_go; in_w = -a_go; }
      else { in_nw = o_nw; in_n = o_n; in_w = o_w; if (l == 0 && on && t >= cw_lo && t <= cw_hi) { in_nw = cs_carry_ld(cNW[t]); in_n = cs_carry_ld(cN[t]); in_w = cs_carry_ld(cW[t]); } }
      const int u_nw = gn_shr1<G>(cur_nw, in_nw, l), u_n = gn_shr1<G>(cur_n, in_n, l), u_w = gn_shr1<G>(cur_w, in_w, l);   // cell_n = (r-1, c)
      const bool inband = on && row_ok && c >= x_min && c <= x_max;
      int n_nw, n_n, n_w;
      {
        const int ms = (db[min(max(c, 0), glen - 1)] == q) ? sc.match : sc.mismatch;
        int tmp, b0, b1, b2, nul = 0;
        if (!revcmpl) {                                            // ref: sw-full-ls.c:264-278
          tmp = d_nw + ms; b0 = 0;
          if (d_n + ms > tmp) { tmp = d_n + ms; b0 = 1; }
          if (d_w + ms > tmp) { tmp = d_w + ms; b0 = 2; }
        } else {                                                   // :279-292
          tmp = d_w + ms; b0 = 2;
          if (d_n + ms > tmp) { tmp = d_n + ms; b0 = 1; }
          if (d_nw + ms > tmp) { tmp = d_nw + ms; b0 = 0; }
        }
        if (LOCAL && tmp <= 0) { tmp = 0; nul |= 0x10; }
        n_nw = tmp;
        if (!revcmpl) {                                            // north :303-320
          tmp = u_nw - b_go - b_ge; b1 = 0;
          if (u_n - b_ge > tmp) { tmp = u_n - b_ge; b1 = 1; }
        } else {
          tmp = u_n - b_ge; b1 = 1;
          if (u_nw - b_go - b_ge > tmp) { tmp = u_nw - b_go - b_ge; b1 = 0; }
        }
        if (LOCAL && tmp <= 0) { tmp = 0; nul |= 0x20; }
        n_n = tmp;
        if (!revcmpl) {                                            // west :330-347
          tmp = pw_nw - a_go - a_ge; b2 = 0;
          if (pw_w - a_ge > tmp) { tmp = pw_w - a_ge; b2 = 1; }
        } else {
          tmp = pw_w - a_ge; b2 = 1;
          if (pw_nw - a_go - a_ge > tmp) { tmp = pw_nw - a_go - a_ge; b2 = 0; }
        }
        if (LOCAL && tmp <= 0) { tmp = 0; nul |= 0x40; }
        n_w = tmp;
        if (inband) back[(size_t)r * glen + c] = (uint8_t)(0x80 | nul | b0 | (b1 << 2) | (b2 << 3));
        if (inband && (LOCAL || last_row_lane)) {                  // :359-368 leftmost strict maximum: of the last read row, or (local) of this lane's rows
          int m = max(n_n, n_nw); m = max(m, n_w);
          if (m > out.score) { out.score = m; out.max_i = r; out.max_j = c; out.e_nw = n_nw; out.e_n = n_n; out.e_w = n_w; }
        }
      }
      if (!inband) { n_nw = o_nw; n_n = o_n; n_w = o_w; }
      if (more && l == G - 1 && on && c >= 0 && c < glen) { cs_carry_st(cNW[c], n_nw); cs_carry_st(cN[c], n_n); cs_carry_st(cW[c], n_w); }
      d_nw = u_nw; d_n = u_n; d_w = u_w;
      pw_nw = n_nw; pw_w = n_w;
      cur_nw = n_nw; cur_n = n_n; cur_w = n_w;
    }
    cw_lo = 1; cw_hi = 0;
    if (more && t_hi >= 0) { cw_lo = max(0, t_lo - (G - 1)); cw_hi = min(glen - 1, t_hi - (G - 1)); }
    if (more) __syncthreads();
  }
  // the group's result: from the last row's lane, or (local) from the lane whose row comes first among those with the largest score
  const int gb = lane & (64 - G);
  int src = (rlen - 1) & (G - 1);
  if (LOCAL) {
    int best = out.score;
    for (int d = G / 2; d > 0; d >>= 1) best = max(best, __shfl_xor(best, d));
    int row = (out.score == best) ? out.max_i : INT_MAX;
    for (int d = G / 2; d > 0; d >>= 1) row = min(row, __shfl_xor(row, d));
    src = (best > 0) ? (row & (G - 1)) : 0;
  }
  out.score = __shfl(out.score, gb | src); out.max_i = __shfl(out.max_i, gb | src); out.max_j = __shfl(out.max_j, gb | src);
  out.e_nw = __shfl(out.e_nw, gb | src); out.e_n = __shfl(out.e_n, gb | src); out.e_w = __shfl(out.e_w, gb | src);
  return out;
}

// The work items of a pass 2 (letter or colour space) listed by kind: forward-strand windows from the front of order[], reverse-strand ones (they take the mirrored tie rules, when
// sc.tiebreak_rev) from its back; cnt[0] / cnt[1] count them, cnt[2] is the unit counter k_pass2_g4 / k_pass2_cs_g4 draw from (all three zeroed by the launch).  The order inside a
// kind is whatever the atomics give: every result is stored under its work index, so the output does not depend on it.
__global__ void __launch_bounds__(256) k_p2cs_classify(const uint32_t* __restrict__ work, const uint32_t* __restrict__ n_work_p, const int32_t* __restrict__ sel,
                                                       int cs_flip, int tiebreak_rev, uint32_t* __restrict__ order, uint32_t* __restrict__ cnt) {
  const uint32_t n_work = *n_work_p;
  const int lane = threadIdx.x & 63;
  for (uint32_t base = (blockIdx.x * 256u + threadIdx.x) & ~63u; base < n_work; base += gridDim.x * 256u) {
    const uint32_t wi = base + (uint32_t)lane;
    bool fw = false, rv = false;
    if (wi < n_work) {
      const uint32_t wk = work[wi];
      const int id = sel[(size_t)(wk >> 6) * SEL_MAX + (wk & 63)];
      rv = ((id >> 16) != cs_flip) && tiebreak_rev; fw = !rv;
    }
    const unsigned long long m_fw = __ballot(fw), m_rv = __ballot(rv);
    uint32_t b_fw = 0, b_rv = 0;
    if (lane == 0) { if (m_fw) b_fw = atomicAdd(&cnt[0], (uint32_t)__popcll(m_fw)); if (m_rv) b_rv = atomicAdd(&cnt[1], (uint32_t)__popcll(m_rv)); }
    b_fw = (uint32_t)__shfl((int)b_fw, 0); b_rv = (uint32_t)__shfl((int)b_rv, 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (fw) order[b_fw + (uint32_t)__popcll(m_fw & below)] = wi;
    if (rv) order[n_work - 1u - (b_rv + (uint32_t)__popcll(m_rv & below))] = wi;
  }
}

template <int G, typename CT, bool LOCAL>
__global__ void __launch_bounds__(GM_WAVE)
k_pass2_g4(GmIndexDev ix, GmScoreDev sc, const uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words,
           GmHit* __restrict__ hits, const uint16_t* __restrict__ perm, int hcap,
           const int32_t* __restrict__ sel, const int32_t* __restrict__ sel_sidx, int input_strand, int write_back,
           const uint32_t* __restrict__ work, const uint32_t* __restrict__ n_work_p,
           GmFullRes* __restrict__ res, uint8_t* __restrict__ ops, int ops_stride,
           uint8_t* __restrict__ back_pool, size_t back_stride, int max_w, unsigned long long* __restrict__ stats,
           const uint32_t* __restrict__ order, uint32_t* __restrict__ cls_cnt) {
  extern __shared__ __align__(16) uint8_t sm[];
  constexpr int NG = 64 / G;
  const int lane = threadIdx.x, g = lane / G, l = lane & (G - 1);
  const int rl16 = (read_len + 15) & ~15, mw16 = (max_w + 15) & ~15;
  uint8_t* qr_all = sm;                                          // NG reads, NG windows, NG sets of carry rows
  uint8_t* db_all = sm + NG * rl16;
  CT* carry_all = (CT*)(db_all + NG * mw16);
  const uint8_t* qr = qr_all + g * rl16; const uint8_t* db = db_all + g * mw16; CT* carry = carry_all + g * 3 * max_w;
  uint8_t* back = back_pool + ((size_t)blockIdx.x * NG + g) * back_stride;
  const uint32_t n_work = *n_work_p;
  unsigned long long vcalls = 0, vcells = 0, fcalls = 0;
  // A pass holds windows of one strand kind (the kind decides the tie rules of a cell, ref: sw-full-ls.c:264-347; as a per-lane value both variants of every update ran, half the
  // lanes masked): k_p2cs_classify has listed the work items by kind -- forward ones from the front of order[], reverse ones from its back -- a pass is one unit of NG list entries
  // of one kind, and the waves draw units from a counter until none is left.
  const uint32_t n_fw = cls_cnt[0], n_rv = cls_cnt[1];
  const uint32_t u_fw = (n_fw + NG - 1) / NG, u_all = u_fw + (n_rv + NG - 1) / NG;
  for (;;) {
    uint32_t u = 0;
    if (lane == 0) u = atomicAdd(&cls_cnt[2], 1u);
    u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
    if (u >= u_all) break;
    const bool rev_u = u >= u_fw;                                  // (wave-uniform)
    const uint32_t idx = (rev_u ? u - u_fw : u) * NG + (uint32_t)g;
    const bool has = idx < (rev_u ? n_rv : n_fw);
    const uint32_t wi = has ? (rev_u ? order[n_work - 1u - idx] : order[idx]) : 0u;
    const uint32_t wk = has ? work[wi] : 0u;
    const int rd = (int)(wk >> 6), k = (int)(wk & 63);
    const int id = has ? sel[(size_t)rd * SEL_MAX + k] : 0;
    int st = id >> 16; const int hi = id & 0xFFFF;
    const size_t slot = ((size_t)rd * 2 + st) * hcap + hi;
    GmHit h; if (has) h = hits[slot]; else { h.g_off = 0; h.ax = h.ay = 0; h.alen = h.awidth = 1; h.score_window_gen = 0; h.score_vector = 0; h.pct_score_vector = 0; h.cn = 0; h.w_len = 1; h.matches = 0; h.flags = 1; }
    const int cn = h.cn, w_len = h.w_len;
    const long long clen = (long long)ix.contig_off[cn + 1] - ix.contig_off[cn];
    long long g_off = h.g_off; long long ax = h.ax, ay = h.ay; int gen_st = 0;
    if (st != input_strand) {                                   // reverse_hit, ref: mapping.c:254-263,337-339; anchor_reverse anchors.h:30-34
      g_off = clen - g_off - w_len;
      ax = -ax + (w_len - 1) - (h.alen - 1) - (h.awidth - 1);
      ay = -ay + (read_len - 1) - (h.alen - 1) + (h.awidth - 1);
      gen_st = 1; st = input_strand;
    }
    const uint64_t g0 = (uint64_t)ix.contig_off[cn] + h.g_off;
    const int rna_bits = ((ix.contig_rna && has && ix.contig_rna[cn]) ? 2 : 0) | ((ix.read_rna && has && ix.read_rna[rd]) ? 4 : 0);
    __syncthreads();
    for (int gg = 0; gg < NG; gg++) {                            // the whole wave unpacks each group's read and window (wave-uniform arguments from the group's first lane)
      if (!__shfl((int)has, gg * G)) continue;
      const int rd_g = __shfl(rd, gg * G), wl_g = __shfl(w_len, gg * G), gs_g = __builtin_amdgcn_readfirstlane(__shfl(gen_st | rna_bits, gg * G));      // (bit 1: the contig is RNA, bit 2: the read is; wave-uniform)
      const uint64_t g0_g = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(g0 >> 32), gg * G) << 32) | (uint32_t)__shfl((int)(uint32_t)g0, gg * G);
      load_read(reads + (size_t)rd_g * read_words, read_len, input_strand != 0, qr_all + gg * rl16, lane, (gs_g & 4) != 0);
      load_window(ix.genome, g0_g, wl_g, (gs_g & 1) != 0, db_all + gg * mw16, lane, (gs_g & 2) != 0);
    }
    __syncthreads();
    const int score_max = (read_len < w_len ? read_len : w_len) * sc.match;
    const int thresh = thr_of(sc.full_thr_frac, sc.full_abs, score_max);
    // re-score only where pass 1 took the value from its cache (see k_pass2)
    int sv = h.score_vector;
    const bool need = has && !(h.flags & 1u);
    for (int gg = 0; gg < NG; gg++) {
      if (!__shfl((int)need, gg * G)) continue;
      __syncthreads();
      const int v = sw_vector_wave(db_all + gg * mw16, __shfl(w_len, gg * G), qr_all + gg * rl16, read_len, sc, (int16_t*)carry_all, lane);
      __syncthreads();
      if (g == gg) { sv = v; if (l == 0) { vcalls++; vcells += (unsigned long long)w_len * read_len; } }
    }
    GmFullRes R;
    R.read_idx = rd; R.st = (int16_t)ix.cs_flip; R.gen_st = (int16_t)gen_st; R.cn = (uint32_t)cn; R.g_off = (uint32_t)g_off; R.w_len = w_len;
    R.score_vector = sv; R.score_max = score_max; R.matches = h.matches; R.score_window_gen = h.score_window_gen;
    R.score = 0; R.read_start = 0; R.rmapped = 0; R.genome_start = 0; R.gmapped = 0;
    R.n_match = R.n_mismatch = R.n_ins = R.n_del = 0; R.n_ops = 0; R.ops_off = (uint32_t)(wi * (uint32_t)ops_stride);
    R.sort_idx = (sel_sidx && has) ? sel_sidx[(size_t)rd * SEL_MAX + k] : 0; R.hit_slot = (uint32_t)slot;
    if (write_back && has && l == 0) hits[slot].score_vector = sv;      // hit_run_full_sw keeps the re-scored value in the hit (ref: mapping.c:386-388)
    const bool act = has && sv >= thresh;
    if (act && l == 0) fcalls++;
    // rectangle = anchor_join(1 anchor) + anchor_widen(anchor_width), ref: sw-full-ls.c:176-178, anchors.c:9-61
    long long rx, ry; int rw, rl;
    { long long nw = ax + ay, sw = ax - ay, ne = sw + 2 * (h.awidth - 1), se = nw + 2 * (h.alen - 1);
      if ((nw + sw) % 2 != 0) nw--;
      rx = (nw + sw) / 2; ry = nw - rx;
      if ((ne - sw) % 2 != 0) ne++;
      rw = (int)((ne - sw) / 2 + 1);
      if ((se - nw) % 2 != 0) se++;
      rl = (int)((se - nw) / 2 + 1);
      rx -= sc.anchor_width / 2; ry += sc.anchor_width / 2; rw += sc.anchor_width; }
    __syncthreads();
    FullOut fo = rev_u ? full_sw_g4<G, CT, LOCAL, true>(db, w_len, qr, read_len, sc, act, rx, ry, rl, rw, back, carry, lane)
                       : full_sw_g4<G, CT, LOCAL, false>(db, w_len, qr, read_len, sc, act, rx, ry, rl, rw, back, carry, lane);
    __syncthreads();
    if (LOCAL) {
      // the filter's best local alignment leaves the anchor band: once more over the band the threshold allows (ref: sw-full-ls.c:395-398)
      const bool again = act && fo.score != sv;
      if (__any(again)) {
        threshold_band(w_len, read_len, sc.match, thresh, &rx, &ry, &rl, &rw);
        const FullOut f2 = rev_u ? full_sw_g4<G, CT, LOCAL, true>(db, w_len, qr, read_len, sc, again, rx, ry, rl, rw, back, carry, lane)
                                 : full_sw_g4<G, CT, LOCAL, false>(db, w_len, qr, read_len, sc, again, rx, ry, rl, rw, back, carry, lane);
        if (again) fo = f2;
        __syncthreads();
      }
    }
    R.score = act ? fo.score : 0;
    if (act && fo.score > 0 && l == 0) {
      // do_backtrace, ref: sw-full-ls.c:413-516 -- the group's first lane walks; ops are emitted reversed then flipped
      int i = fo.max_i, j = fo.max_j;
      int state = 0, fs = fo.e_nw;                               // from-state: 0 nw, 1 n, 2 w  (ref :420-427: nw, then w if strictly greater, then n if strictly greater)
      if (fo.e_w > fs) { state = 2; fs = fo.e_w; }
      if (fo.e_n > fs) state = 1;
      uint8_t* o = ops + (size_t)R.ops_off;
      int no = 0, rstart = 0, gstart = 0, nm = 0, nmm = 0, nin = 0, ndel = 0;
      while (i >= 0 && j >= 0) {
        const uint8_t bb = __hip_atomic_load(&back[(size_t)i * w_len + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(bb & 0x80)) break;                // out-of-band cell: back == 0 in the reference
        if (LOCAL) {                            // a floored state has a null back pointer; a cell outside the band was never computed (stale byte)
          if ((bb >> (4 + state)) & 1) break;
          int bx_min, bx_max; band_range(rx, ry, rl, rw, w_len, i, &bx_min, &bx_max);
          if (j < bx_min || j > bx_max) break;
        }
        int nstate;
        if (state == 1) {                       // FROM_NORTH_*: BACK_DELETION (gap in the genome)
          if (no < ops_stride) o[no] = 'D'; no++; ndel++; rstart = i; i--;
          nstate = ((bb >> 2) & 1) ? 1 : 0;
        } else if (state == 2) {                // FROM_WEST_*: BACK_INSERTION (gap in the read)
          if (no < ops_stride) o[no] = 'I'; no++; nin++; gstart = j; j--;
          nstate = ((bb >> 3) & 1) ? 2 : 0;
        } else {                                // FROM_NORTHWEST_*
          if (no < ops_stride) o[no] = 'M'; no++;
          if (db[j] == qr[i]) nm++; else nmm++;
          rstart = i; gstart = j; i--; j--;
          nstate = (bb & 3);                    // 0 nw, 1 n, 2 w
        }
        state = nstate;
      }
      const int nov = min(no, ops_stride);
      for (int a = 0, b = nov - 1; a < b; a++, b--) { uint8_t tt = o[a]; o[a] = o[b]; o[b] = tt; }
      R.n_ops = no; R.read_start = rstart; R.genome_start = gstart + (int)g_off;
      R.gmapped = fo.max_j - gstart + 1; R.rmapped = fo.max_i - rstart + 1;
      R.n_match = nm; R.n_mismatch = nmm; R.n_ins = nin; R.n_del = ndel;
    }
    if (has && l == 0) res[wi] = R;
  }
  for (int d = 32; d >= G; d >>= 1) { fcalls += __shfl_xor(fcalls, d); vcalls += __shfl_xor(vcalls, d); vcells += __shfl_xor(vcells, d); }
  if (lane == 0) { GS_ADD(stats, GS_FULL_CALLS, fcalls); GS_ADD(stats, GS_VEC_CALLS, vcalls); GS_ADD(stats, GS_VEC_CELLS, vcells); }
}

// work list = (read << 6 | k) for every selected hit, built by a scan-free atomic append (order fixed up on the host by key)
__global__ void __launch_bounds__(256) k_build_work(int n_reads, const uint32_t* __restrict__ sel_cnt, const uint32_t* __restrict__ sel_off,
                                                     uint32_t* __restrict__ work) {
  const int rd = blockIdx.x * blockDim.x + threadIdx.x;
  if (rd >= n_reads) return;
  const uint32_t n = sel_cnt[rd], o = sel_off[rd];
  for (uint32_t k = 0; k < n; k++) work[o + k] = ((uint32_t)rd << 6) | k;
}

// exclusive scan of sel_cnt (single block; n_reads per sub-batch is modest) -> sel_off, total in n_work
__global__ void __launch_bounds__(1024) k_scan_counts(int n, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ off, uint32_t* __restrict__ total) {
  __shared__ uint32_t part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int a0 = tid * per, a1 = min(n, a0 + per);
  uint32_t s = 0;
  for (int a = a0; a < a1; a++) s += cnt[a];
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) { uint32_t v = (tid >= d) ? part[tid - d] : 0; __syncthreads(); part[tid] += v; __syncthreads(); }
  uint32_t run = part[tid] - s;
  for (int a = a0; a < a1; a++) { off[a] = run; run += cnt[a]; }
  if (tid == 1023) *total = part[1023];
}

// ---------------------------------------------------------------------------------------------
// S1 batch: n independent (window, read) pairs on caller bitfields
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GM_WAVE)
k_sw_vector_batch(GmScoreDev sc, int n, const uint32_t* __restrict__ genome, const long long* __restrict__ goff, const int* __restrict__ glen,
                  const uint32_t* __restrict__ reads, int read_words, const int* __restrict__ rlen, int max_g, int max_r, int* __restrict__ scores,
                  int early_thr, uint8_t* __restrict__ stopped) {
  extern __shared__ __align__(16) uint8_t sm[];
  const int lane = threadIdx.x;
  uint8_t* qr = sm;
  uint8_t* db = sm + ((max_r + 15) & ~15);
  int16_t* carry = (int16_t*)(db + ((max_g + 15) & ~15));
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    __syncthreads();
    load_read(reads + (size_t)i * read_words, rlen[i], false, qr, lane);
    load_window(genome, (uint64_t)goff[i], glen[i], false, db, lane);
    __syncthreads();
    bool cut = false;
    const int s = early_thr > 0 ? sw_vector_wave_t<false>(db, nullptr, glen[i], qr, rlen[i], sc, carry, lane, early_thr, &cut) : sw_vector_wave(db, glen[i], qr, rlen[i], sc, carry, lane);
    if (lane == 0) { scores[i] = s; if (stopped) stopped[i] = cut ? 1 : 0; }
  }
}

// ---------------------------------------------------------------------------------------------
// S2 single call: sw_full_ls on caller bitfields (ref: common/sw-full-ls.c:637-683), global mode.
// out[0..12] = score read_start rmapped genome_start gmapped matches mismatches insertions deletions n_ops max_i max_j 0
// ---------------------------------------------------------------------------------------------
template <bool LOCAL>
__global__ void __launch_bounds__(GM_WAVE)
k_sw_full_single(GmScoreDev sc, const uint32_t* __restrict__ genome, long long goff, int glen, const uint32_t* __restrict__ read, int rlen,
                 long long ax, long long ay, int alen, int awidth, int has_anchor, int thresh, int maxscore, int revcmpl,
                 uint8_t* __restrict__ back, int* __restrict__ out, uint8_t* __restrict__ ops, int ops_cap) {
  extern __shared__ __align__(16) uint8_t sm[];
  const int lane = threadIdx.x;
  uint8_t* qr = sm;
  uint8_t* db = sm + ((rlen + 15) & ~15);
  int* carry = (int*)(db + ((glen + 15) & ~15));
  load_read(read, rlen, false, qr, lane);
  load_window(genome, (uint64_t)goff, glen, false, db, lane);
  __syncthreads();
  long long rx, ry; int rw, rl;
  if (has_anchor) {
    long long nw = ax + ay, sw = ax - ay, ne = sw + 2 * (awidth - 1), se = nw + 2 * (alen - 1);
    if ((nw + sw) % 2 != 0) nw--;
    rx = (nw + sw) / 2; ry = nw - rx;
    if ((ne - sw) % 2 != 0) ne++;
    rw = (int)((ne - sw) / 2 + 1);
    if ((se - nw) % 2 != 0) se++;
    rl = (int)((se - nw) / 2 + 1);
    rx -= sc.anchor_width / 2; ry += sc.anchor_width / 2; rw += sc.anchor_width;
  } else threshold_band(glen, rlen, sc.match, thresh, &rx, &ry, &rl, &rw);       // ref: sw-full-ls.c:179-192
  FullOut fo = full_sw_wave<LOCAL>(db, glen, qr, rlen, sc, revcmpl != 0, rx, ry, rl, rw, back, carry, lane);
  __syncthreads();
  if (LOCAL && has_anchor && fo.score != maxscore) {                              // ref: sw-full-ls.c:395-398
    threshold_band(glen, rlen, sc.match, thresh, &rx, &ry, &rl, &rw);
    fo = full_sw_wave<LOCAL>(db, glen, qr, rlen, sc, revcmpl != 0, rx, ry, rl, rw, back, carry, lane);
    __syncthreads();
  }
  if (lane == 0) {
    int i = fo.max_i, j = fo.max_j, no = 0, rstart = 0, gstart = 0, nm = 0, nmm = 0, nin = 0, ndel = 0;
    if (fo.score > 0) {
      int state = 0, fs = fo.e_nw;
      if (fo.e_w > fs) { state = 2; fs = fo.e_w; }
      if (fo.e_n > fs) state = 1;
      while (i >= 0 && j >= 0) {
        const uint8_t bb = __hip_atomic_load(&back[(size_t)i * glen + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(bb & 0x80)) break;
        if (LOCAL) {
          if ((bb >> (4 + state)) & 1) break;
          int bx_min, bx_max; band_range(rx, ry, rl, rw, glen, i, &bx_min, &bx_max);
          if (j < bx_min || j > bx_max) break;
        }
        int nstate;
        if (state == 1) { if (no < ops_cap) ops[no] = 'D'; no++; ndel++; rstart = i; i--; nstate = ((bb >> 2) & 1) ? 1 : 0; }
        else if (state == 2) { if (no < ops_cap) ops[no] = 'I'; no++; nin++; gstart = j; j--; nstate = ((bb >> 3) & 1) ? 2 : 0; }
        else { if (no < ops_cap) ops[no] = 'M'; no++; if (db[j] == qr[i]) nm++; else nmm++; rstart = i; gstart = j; i--; j--; nstate = (bb & 3); }
        state = nstate;
      }
      const int nov = min(no, ops_cap);
      for (int a = 0, b = nov - 1; a < b; a++, b--) { uint8_t tt = ops[a]; ops[a] = ops[b]; ops[b] = tt; }
    }
    out[0] = fo.score; out[1] = rstart; out[2] = fo.max_i - rstart + 1; out[3] = gstart + (int)goff; out[4] = fo.max_j - gstart + 1;
    out[5] = nm; out[6] = nmm; out[7] = nin; out[8] = ndel; out[9] = no; out[10] = fo.max_i; out[11] = fo.max_j;
  }
}

int gm_launch_sw_full_single(const GmScoreDev& sc, const uint32_t* d_genome, long long goff, int glen, const uint32_t* d_read, int rlen,
                             long long ax, long long ay, int alen, int awidth, int revcmpl, uint8_t* d_back, int* d_out, uint8_t* d_ops, int ops_cap,
                             hipStream_t stream, int has_anchor, int thresh, int maxscore, int local) {
  const size_t lds = ((rlen + 15) & ~15) + ((glen + 15) & ~15) + (size_t)glen * 12 + 64;
  if (local)
    hipLaunchKernelGGL(k_sw_full_single<true>, dim3(1), dim3(GM_WAVE), lds, stream, sc, d_genome, goff, glen, d_read, rlen, ax, ay, alen, awidth, has_anchor, thresh,
                       maxscore, revcmpl, d_back, d_out, d_ops, ops_cap);
  else
    hipLaunchKernelGGL(k_sw_full_single<false>, dim3(1), dim3(GM_WAVE), lds, stream, sc, d_genome, goff, glen, d_read, rlen, ax, ay, alen, awidth, has_anchor, thresh,
                       maxscore, revcmpl, d_back, d_out, d_ops, ops_cap);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

// ---- launchers ---------------------------------------------------------------------------------
int gm_launch_pass1(const GmIndexDev& ix, const GmScoreDev& sc, const uint32_t* d_reads, int n_reads, int read_len, int read_words,
                    int window_len, int window_overlap_abs, GmHit* d_hits, const uint16_t* d_perm, const uint32_t* d_hit_cnt, int hcap,
                    unsigned long long* d_slots, unsigned long long* d_stats, hipStream_t stream, const int32_t* d_pair_min, const uint8_t* d_saved,
                    const uint8_t* d_initbp, bool early_stop) {
  if (n_reads == 0) return GM_OK;
  // The carry rows of the vector filter exist only for reads of more than 128 bases (two stripes): without them a wave's LDS is 320 B at 100 bp instead of 880 B,
  // and LDS is what bounds the filter's waves beside the seed lookup (k_lookup_v5 leaves ~5 KB of a CU's 160 KB: 5 waves at 1 KB each, 10 at 512 B).
  const size_t carry_bytes = (read_len > 128 && window_len > 256) ? (size_t)window_len * 4 : 0;      // (windows of up to 256 columns carry the stripe's last row in registers)
  // the early stop of a window that cannot reach the threshold (sw_vector_wave_s): only where a score below the threshold is never read again (unpaired
  // reads: the caller says so), with a scoring scheme in which a cell gains at most `match` and a gap never gains, and within the 16-bit range of the test
  const int early = (early_stop && !d_pair_min && !sc.gapless && sc.match > 0 && sc.mismatch <= sc.match && sc.a_go >= 0 && sc.a_ge >= 0 && sc.b_go >= 0 && sc.b_ge >= 0 &&
                     sc.match * (2 * 128 + 2) < 32000 && hcap <= 65536 &&
                     !(gm_tune("GM_P1_EARLY") && atoi(gm_tune("GM_P1_EARLY")) == 0)) ? 1 : 0;
  if (ix.colour) {
    if (!d_initbp) { gm_set_error("pass 1 in colour space needs the primer letters"); return GM_E_ARG; }
    const size_t lds = ((read_len + 15) & ~15) + 2 * (size_t)((window_len + 15) & ~15) + carry_bytes + 64;
    hipLaunchKernelGGL(k_pass1<true>, dim3(n_reads * 2), dim3(GM_WAVE), lds, stream, ix, sc, d_reads, n_reads, read_len, read_words,
                       window_len, window_overlap_abs, d_hits, d_perm, d_hit_cnt, hcap, d_slots, d_stats, d_pair_min, d_saved, d_initbp, early);
  } else {
    const size_t lds = ((read_len + 15) & ~15) + ((window_len + 15) & ~15) + carry_bytes + 64;
    hipLaunchKernelGGL(k_pass1<false>, dim3(n_reads * 2), dim3(GM_WAVE), lds, stream, ix, sc, d_reads, n_reads, read_len, read_words,
                       window_len, window_overlap_abs, d_hits, d_perm, d_hit_cnt, hcap, d_slots, d_stats, d_pair_min, d_saved, (const uint8_t*)nullptr, early);
  }
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_launch_select(const GmScoreDev& sc, int n_reads, int read_len, const GmHit* d_hits, const uint16_t* d_perm,
                     const uint32_t* d_hit_cnt, int hcap, int32_t* d_sel, uint32_t* d_sel_cnt, uint32_t* d_sel_off,
                     uint32_t* d_work, uint32_t* d_n_work, hipStream_t stream, const uint8_t* d_saved) {
  if (n_reads == 0) return GM_OK;
  hipLaunchKernelGGL(k_select, dim3((n_reads + 63) / 64), dim3(64), 0, stream, sc, n_reads, read_len, d_hits, d_perm, d_hit_cnt, hcap, d_sel, d_sel_cnt, d_saved);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, stream, n_reads, d_sel_cnt, d_sel_off, d_n_work);
  hipLaunchKernelGGL(k_build_work, dim3((n_reads + 255) / 256), dim3(256), 0, stream, n_reads, d_sel_cnt, d_sel_off, d_work);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_launch_build_work(int n_reads, const uint32_t* d_sel_cnt, uint32_t* d_sel_off, uint32_t* d_work, uint32_t* d_n_work, hipStream_t stream) {
  if (n_reads == 0) return GM_OK;
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, stream, n_reads, d_sel_cnt, d_sel_off, d_n_work);
  hipLaunchKernelGGL(k_build_work, dim3((n_reads + 255) / 256), dim3(256), 0, stream, n_reads, d_sel_cnt, d_sel_off, d_work);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_launch_pass2(const GmIndexDev& ix, const GmScoreDev& sc, const uint32_t* d_reads, int n_reads, int read_len, int read_words,
                    int window_len, GmHit* d_hits, const uint16_t* d_perm, int hcap, const int32_t* d_sel, const uint32_t* d_sel_cnt,
                    const uint32_t* d_work, const uint32_t* d_n_work, GmFullRes* d_res, uint8_t* d_ops, int ops_stride,
                    uint8_t* d_back, size_t back_stride, int grid, unsigned long long* d_stats, hipStream_t stream,
                    const int32_t* d_sel_sidx, int input_strand, int write_back, uint32_t* d_order, uint32_t* d_cls_cnt) {
  if (n_reads == 0) return GM_OK;
  const int p2_ablate = gm_tune("GM_P2_ABLATE") ? atoi(gm_tune("GM_P2_ABLATE")) : 0;
  size_t lds = ((read_len + 15) & ~15) + ((window_len + 15) & ~15) + (size_t)window_len * 12 + 64;
  const size_t back_bytes = (size_t)read_len * window_len;
#define GM_P2_LAUNCH(BL, LOC) hipLaunchKernelGGL((k_pass2<BL, LOC>), dim3(grid), dim3(GM_WAVE), lds, stream, ix, sc, d_reads, n_reads, read_len, read_words, \
    d_hits, d_perm, hcap, d_sel, d_sel_sidx, input_strand, write_back, d_sel_cnt, d_work, d_n_work, d_res, d_ops, ops_stride, d_back, back_stride, window_len, d_stats, p2_ablate)
  // Back pointers in LDS (14 KB per wave at 100 bp) cap the CU at ten waves; in a per-wave global scratch (L2-resident) the CU runs at full
  // occupancy: measured 109 -> 37 ms per 1 M reads on the 3 Gbp workload.  The LDS form stays for comparison (GM_P2_BACK_LDS=1).
  // Four windows per wave (k_pass2_g4) unless GM_P2_G4=0 asks for the one-window kernel; its wave owns that many consecutive back-pointer scratches.  Eight windows (groups of
  // 8 lanes, int16_t carry rows, see cs_carry_ld) were built and measured for letter space too (tuning builds: GM_P2_G=8): 18.9 against 15.8 ms per 1 M 100-base reads -- the
  // letter-space cell is ~30 instructions, so the thirteen stripes' set-up and the carry traffic of every step weigh more than the steps saved; colour space (360 instructions
  // a cell) takes eight.
  const bool g4 = !(gm_tune("GM_P2_G4") && atoi(gm_tune("GM_P2_G4")) == 0) && !gm_tune("GM_P2_BACK_LDS") && grid >= 8 && d_order && d_cls_cnt;
  if (g4) {
    const size_t r16 = (size_t)((read_len + 15) & ~15), w16 = (size_t)((window_len + 15) & ~15);
    // no score on a path through the matrix lies further from 0 than `reach`: a path has at most read_len + window_len moves, and a move changes the score by a match, a
    // mismatch or a gap's opening + extension at most
    const int big = std::max(std::max(std::abs(sc.match), std::abs(sc.mismatch)), std::max(std::max(std::abs(sc.a_go) + std::abs(sc.a_ge), std::abs(sc.b_go) + std::abs(sc.b_ge)), 1));
    const long long reach = (long long)(read_len + window_len) * big;
    const size_t lds8 = 8 * r16 + 8 * w16 + 8 * (size_t)window_len * 3 * sizeof(int16_t) + 64;
    const size_t lds4 = 4 * r16 + 4 * w16 + 4 * (size_t)window_len * 3 * sizeof(int) + 64;
#ifdef GM_TUNING
    const bool g8 = reach < 16000 && lds8 <= 64 * 1024 && gm_tune("GM_P2_G") && atoi(gm_tune("GM_P2_G")) == 8;
#else
    const bool g8 = false; (void)reach; (void)lds8;
#endif
    const size_t ldsn = g8 ? lds8 : lds4;
    static GmLdsLimit lim4; size_t& conf4 = lim4.cur();
    if (ldsn > 48 * 1024 && ldsn > conf4) {
      GM_HIP(hipFuncSetAttribute((const void*)k_pass2_g4<16, int, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
      GM_HIP(hipFuncSetAttribute((const void*)k_pass2_g4<16, int, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
#ifdef GM_TUNING
      GM_HIP(hipFuncSetAttribute((const void*)k_pass2_g4<8, int16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
      GM_HIP(hipFuncSetAttribute((const void*)k_pass2_g4<8, int16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
#endif
      conf4 = ldsn; }
    GM_HIP(hipMemsetAsync(d_cls_cnt, 0, 16, stream));
    hipLaunchKernelGGL(k_p2cs_classify, dim3(512), dim3(256), 0, stream, d_work, d_n_work, d_sel, input_strand, sc.tiebreak_rev ? 1 : 0, d_order, d_cls_cnt);
#define GM_P2_G4L(GG, CT, LOC) hipLaunchKernelGGL((k_pass2_g4<GG, CT, LOC>), dim3(grid / (64 / GG)), dim3(GM_WAVE), ldsn, stream, ix, sc, d_reads, n_reads, read_len, read_words, d_hits, d_perm, hcap, d_sel, d_sel_sidx, \
                                     input_strand, write_back, d_work, d_n_work, d_res, d_ops, ops_stride, d_back, back_stride, window_len, d_stats, d_order, d_cls_cnt)
#ifdef GM_TUNING
    if (g8) { if (sc.local) GM_P2_G4L(8, int16_t, true); else GM_P2_G4L(8, int16_t, false); } else
#endif
    { if (sc.local) GM_P2_G4L(16, int, true); else GM_P2_G4L(16, int, false); }
#undef GM_P2_G4L
  } else
  if (back_bytes <= 40 * 1024 && gm_tune("GM_P2_BACK_LDS")) {
    lds += back_bytes + 16;
    if (sc.local) GM_P2_LAUNCH(true, true); else GM_P2_LAUNCH(true, false);
  } else {
    if (sc.local) GM_P2_LAUNCH(false, true); else GM_P2_LAUNCH(false, false);
  }
#undef GM_P2_LAUNCH
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_launch_sw_vector_batch(const GmScoreDev& sc, int n, const uint32_t* d_genome, const long long* d_goff, const int* d_glen,
                              const uint32_t* d_reads, int read_words, const int* d_rlen, int max_g, int max_r, int* d_scores, hipStream_t stream,
                              int early_thr, uint8_t* d_stopped) {
  if (n == 0) return GM_OK;
  const size_t lds = ((max_r + 15) & ~15) + ((max_g + 15) & ~15) + (size_t)max_g * 4 + 64;
  const int grid = std::min(n, 256 * 16);
  hipLaunchKernelGGL(k_sw_vector_batch, dim3(grid), dim3(GM_WAVE), lds, stream, sc, n, d_genome, d_goff, d_glen, d_reads, read_words, d_rlen, max_g, max_r, d_scores,
                     early_thr, d_stopped);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

// =============================================================================================
// Colour space (S1/S2 seams; the CS read pipeline around them is not built yet)
// =============================================================================================
__global__ void __launch_bounds__(GM_WAVE)
k_sw_vector_batch_cs(GmScoreDev sc, int n, const uint32_t* __restrict__ genome_cs, const uint32_t* __restrict__ genome_ls,
                     const long long* __restrict__ goff, const int* __restrict__ glen, const uint32_t* __restrict__ reads, int read_words,
                     const int* __restrict__ rlen, const int* __restrict__ initbp, int max_g, int max_r, int* __restrict__ scores) {
  extern __shared__ __align__(16) uint8_t sm[];
  const int lane = threadIdx.x;
  uint8_t* qr = sm;
  uint8_t* db = sm + ((max_r + 15) & ~15);
  uint8_t* db0 = db + ((max_g + 15) & ~15);
  int16_t* carry = (int16_t*)(db0 + ((max_g + 15) & ~15));
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    __syncthreads();
    load_read(reads + (size_t)i * read_words, rlen[i], false, qr, lane);
    load_window(genome_cs, (uint64_t)goff[i], glen[i], false, db, lane);
    load_window(genome_ls, (uint64_t)goff[i], glen[i], false, db0, lane);
    __syncthreads();
    for (int c = lane; c < glen[i]; c += GM_WAVE) db0[c] = (uint8_t)cs_lstocs(db0[c], initbp[i] & 0xff, (initbp[i] & GM_SEAM_RNA) != 0);     // first-colour row, ref: sw-vector.c:131 (is_rna rides in bit 8 of the primer word)
    __syncthreads();
    const int s = sw_vector_wave_t<true>(db, db0, glen[i], qr, rlen[i], sc, carry, lane);
    if (lane == 0) scores[i] = s;
  }
}

struct GmCsDev { int match, mismatch, xover, a_go, a_ge, b_go, b_ge, anchor_width, taboo; };
#ifdef GM_BAND_DEBUG
__device__ unsigned long long g_band_dbg[8];
extern "C" int gm_debug_band(unsigned long long* out) { unsigned long long z[8] = {0}; if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_band_dbg), sizeof z) != hipSuccess) return -1; return hipMemcpyToSymbol(HIP_SYMBOL(g_band_dbg), z, sizeof z) == hipSuccess ? 0 : -1; }
#endif

// sw_full_cs (ref: common/sw-full-cs.c:249-623): lane = read row, column t - lane at step t, twelve running values per cell
// (4 layers x {nw, n, w}); what row r needs from row r - 1 arrives by DPP shifts.  back[cell] = three words of four codes
// (dir << 2 | layer) for the nw / n / w states of the four layers.  Out-of-band cells are -INT_MAX/2 as init_cell(.., 0, ..) leaves them.
struct CsBest { int score, i, j, k, e_nw, e_n, e_w; };
// xrow: per-position crossover scores of this read (from its QVs, ref: gmapper.c:532-544) or null = the global one everywhere
// REV / TABOO as template constants: the tie rules of a reverse-strand window and the indel-taboo tests sit inside every one of the 36 state updates of
// a step; as run-time values they became a uniform branch (with its register copies) each -- two thirds of the step's scalar instructions.
template <bool REV, bool TABOO>
__device__ CsBest full_sw_cs_wave_t(const uint8_t* db, int glen, const uint8_t* qr4, int qstride, int rlen, const GmCsDev& P,
                                    long long rx, long long ry, int rl, int rw, uint32_t* back, int* carry, int lane, const int8_t* xrow = nullptr) {
  constexpr bool revcmpl = REV;
  CsBest best; best.score = 0; best.i = best.j = best.k = 0; best.e_nw = best.e_n = best.e_w = 0;
  const int xg = P.xover;                                 // global_xover_penalty: the virtual row above the matrix (ref: sw-full-cs.c:270)
  int xo = xg;
  const int n_stripes = (rlen + 63) >> 6;
  int cw_lo = 1, cw_hi = 0;                              // columns of the carry rows the previous stripe wrote
  for (int s = 0; s < n_stripes; s++) {
    const int r = s * 64 + lane;
    const bool row_ok = r < rlen;
    if (xrow) xo = row_ok ? (int)xrow[r] : xg;           // ref: sw-full-cs.c:312
    int q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = row_ok ? qr4[k * qstride + r] : 0x7F;
    int x_min = 0, x_max = -1;
    if (row_ok) band_range(rx, ry, rl, rw, glen, r, &x_min, &x_max);
    const bool notaboo = TABOO ? r < rlen - P.taboo : true;
    // only the steps at which some lane stands inside the band (see full_sw_wave)
    int t_lo = INT_MAX, t_hi = -1;
    if (row_ok && x_max >= x_min) { t_lo = x_min + lane; t_hi = x_max + lane; }
    for (int dd = 32; dd > 0; dd >>= 1) { t_lo = min(t_lo, __shfl_xor(t_lo, dd)); t_hi = max(t_hi, __shfl_xor(t_hi, dd)); }
#ifdef GM_BAND_DEBUG
    if (lane == 0) { atomicAdd(&g_band_dbg[0], (unsigned long long)(glen + min(64, rlen - s * 64) - 1)); atomicAdd(&g_band_dbg[1], (unsigned long long)max(0, t_hi - t_lo + 1)); atomicAdd(&g_band_dbg[2], 1ull);
                     atomicAdd(&g_band_dbg[3], (unsigned long long)rw); atomicAdd(&g_band_dbg[4], (unsigned long long)rl); }
#endif
    int pw[12], d[12], cur[12];                      // (r, c-1), (r-1, c-1), this lane's latest cell; index = layer * 3 + {0 nw, 1 n, 2 w}
#pragma unroll
    for (int x = 0; x < 12; x++) { pw[x] = FS_NEG; d[x] = FS_NEG; cur[x] = FS_NEG; }
    if (lane == 0) {
      if (s == 0) {                                  // virtual row -1 (every column of it, -1 included): init_cell(.., 1, xover), ref :201-215
#pragma unroll
        for (int k = 0; k < 4; k++) { const int x = k ? xg : 0; d[k * 3] = x; d[k * 3 + 1] = -P.b_go + x; d[k * 3 + 2] = -P.a_go + x; }
      } else if (t_hi >= 0 && t_lo >= 1 && t_lo - 1 >= cw_lo && t_lo - 1 <= cw_hi) {   // (r-1, t_lo-1) of the previous stripe's last row
#pragma unroll
        for (int x = 0; x < 12; x++) d[x] = carry[x * glen + t_lo - 1];
      }
    }
    const bool more = (s + 1 < n_stripes);
    const bool last_row_lane = row_ok && (r == rlen - 1);
    for (int t = t_lo; t <= t_hi; t++) {
      const int c = t - lane;
      int u[12], inv[12];
      // lane 0's upper neighbour: the virtual row (stripe 0: constants, no memory access at all), or the previous stripe's last row -- one wave-uniform
      // branch around the twelve loads (per-value conditions cost a masked load and a wait each, every step, also in the single-stripe case)
#pragma unroll
      for (int x = 0; x < 12; x++) { const int k = x / 3, st = x % 3; const int xv = k ? xg : 0; inv[x] = s == 0 ? (st == 0 ? 0 : (st == 1 ? -P.b_go : -P.a_go)) + xv : FS_NEG; }
      if (__builtin_amdgcn_readfirstlane((int)(s > 0 && t >= cw_lo && t <= cw_hi))) {
#pragma unroll
        for (int x = 0; x < 12; x++) inv[x] = carry[x * glen + t];
      }
#pragma unroll
      for (int x = 0; x < 12; x++) u[x] = shr1_i(cur[x], inv[x]);   // cell (r-1, c)
      const bool inband = row_ok && c >= x_min && c <= x_max;
      int nv[12];
#pragma unroll
      for (int x = 0; x < 12; x++) nv[x] = FS_NEG;
      if (inband) {
        const int dbc = db[c];
        uint32_t bw_nw = 0, bw_n = 0, bw_w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int ms = (dbc == 15 || q[k] == 15) ? 0 : (dbc == q[k] ? P.match : P.mismatch);
          int tmp, b;
          // northwest, ref :356-438
          if (!revcmpl) {
            tmp = d[k * 3] + ms; b = (6 << 2) | k;
            if (notaboo && d[k * 3 + 1] + ms > tmp) { tmp = d[k * 3 + 1] + ms; b = (5 << 2) | k; }
            if (d[k * 3 + 2] + ms > tmp) { tmp = d[k * 3 + 2] + ms; b = (7 << 2) | k; }
          } else {
            tmp = d[k * 3 + 2] + ms; b = (7 << 2) | k;
            if (notaboo && d[k * 3 + 1] + ms > tmp) { tmp = d[k * 3 + 1] + ms; b = (5 << 2) | k; }
            if (d[k * 3] + ms > tmp) { tmp = d[k * 3] + ms; b = (6 << 2) | k; }
          }
#pragma unroll
          for (int l = 0; l < 4; l++) {
            if (l == k) continue;
            if (!revcmpl) {
              if (d[l * 3] + ms + xo > tmp) { tmp = d[l * 3] + ms + xo; b = (6 << 2) | l; }
              if (notaboo && d[l * 3 + 1] + ms + xo > tmp) { tmp = d[l * 3 + 1] + ms + xo; b = (5 << 2) | l; }
              if (d[l * 3 + 2] + ms + xo > tmp) { tmp = d[l * 3 + 2] + ms + xo; b = (7 << 2) | l; }
            } else {
              if (d[l * 3 + 2] + ms + xo > tmp) { tmp = d[l * 3 + 2] + ms + xo; b = (7 << 2) | l; }
              if (notaboo && d[l * 3 + 1] + ms + xo > tmp) { tmp = d[l * 3 + 1] + ms + xo; b = (5 << 2) | l; }
              if (d[l * 3] + ms + xo > tmp) { tmp = d[l * 3] + ms + xo; b = (6 << 2) | l; }
            }
          }
          nv[k * 3] = tmp; bw_nw |= (uint32_t)b << (8 * k);
          // north, ref :447-503
          if (!revcmpl) {
            tmp = u[k * 3] - P.b_go - P.b_ge; b = (2 << 2) | k;
            if (!notaboo || u[k * 3 + 1] - P.b_ge > tmp) { tmp = u[k * 3 + 1] - P.b_ge; b = (1 << 2) | k; }
          } else {
            tmp = u[k * 3 + 1] - P.b_ge; b = (1 << 2) | k;
            if (notaboo && u[k * 3] - P.b_go - P.b_ge > tmp) { tmp = u[k * 3] - P.b_go - P.b_ge; b = (2 << 2) | k; }
          }
#pragma unroll
          for (int l = 0; l < 4; l++) {
            if (l == k) continue;
            if (!revcmpl) {
              if (notaboo && u[l * 3] - P.b_go - P.b_ge + xo > tmp) { tmp = u[l * 3] - P.b_go - P.b_ge + xo; b = (2 << 2) | l; }
              if (u[l * 3 + 1] - P.b_ge + xo > tmp) { tmp = u[l * 3 + 1] - P.b_ge + xo; b = (1 << 2) | l; }
            } else {
              if (u[l * 3 + 1] - P.b_ge + xo > tmp) { tmp = u[l * 3 + 1] - P.b_ge + xo; b = (1 << 2) | l; }
              if (notaboo && u[l * 3] - P.b_go - P.b_ge + xo > tmp) { tmp = u[l * 3] - P.b_go - P.b_ge + xo; b = (2 << 2) | l; }
            }
          }
          nv[k * 3 + 1] = tmp; bw_n |= (uint32_t)b << (8 * k);
          // west, ref :512-541 (no crossover on a genomic gap)
          if (!revcmpl) {
            tmp = pw[k * 3] - P.a_go - P.a_ge; b = (3 << 2) | k;
            if (!notaboo || pw[k * 3 + 2] - P.a_ge > tmp) { tmp = pw[k * 3 + 2] - P.a_ge; b = (4 << 2) | k; }
          } else {
            tmp = pw[k * 3 + 2] - P.a_ge; b = (4 << 2) | k;
            if (notaboo && pw[k * 3] - P.a_go - P.a_ge > tmp) { tmp = pw[k * 3] - P.a_go - P.a_ge; b = (3 << 2) | k; }
          }
          nv[k * 3 + 2] = tmp; bw_w |= (uint32_t)b << (8 * k);
          if (last_row_lane) {                         // ref :547-575
            const int a0 = revcmpl ? nv[k * 3 + 2] : nv[k * 3], a1 = nv[k * 3 + 1], a2 = revcmpl ? nv[k * 3] : nv[k * 3 + 2];
            const int m = max(a0, max(a1, a2));
            if (m > best.score) { best.score = m; best.i = r; best.j = c; best.k = k; best.e_nw = nv[k * 3]; best.e_n = nv[k * 3 + 1]; best.e_w = nv[k * 3 + 2]; }
          }
        }
        uint32_t* bp = back + ((size_t)r * glen + c) * 3;
        bp[0] = bw_nw; bp[1] = bw_n; bp[2] = bw_w;
      }
      if (more && lane == 63 && c >= 0 && c < glen) {
#pragma unroll
        for (int x = 0; x < 12; x++) carry[x * glen + c] = nv[x];
      }
#pragma unroll
      for (int x = 0; x < 12; x++) { d[x] = u[x]; pw[x] = nv[x]; cur[x] = nv[x]; }
    }
    cw_lo = 1; cw_hi = 0;
    if (more && t_hi >= 0) { cw_lo = max(0, t_lo - 63); cw_hi = min(glen - 1, t_hi - 63); }
    if (more) __syncthreads();
  }
  const int src = (rlen - 1) & 63;
  best.score = __shfl(best.score, src); best.i = __shfl(best.i, src); best.j = __shfl(best.j, src); best.k = __shfl(best.k, src);
  best.e_nw = __shfl(best.e_nw, src); best.e_n = __shfl(best.e_n, src); best.e_w = __shfl(best.e_w, src);
  return best;
}
template <bool TABOO>
__device__ CsBest full_sw_cs_wave(const uint8_t* db, int glen, const uint8_t* qr4, int qstride, int rlen, const GmCsDev& P, bool revcmpl,
                                  long long rx, long long ry, int rl, int rw, uint32_t* back, int* carry, int lane, const int8_t* xrow = nullptr) {
  return revcmpl ? full_sw_cs_wave_t<true, TABOO>(db, glen, qr4, qstride, rlen, P, rx, ry, rl, rw, back, carry, lane, xrow)
                 : full_sw_cs_wave_t<false, TABOO>(db, glen, qr4, qstride, rlen, P, rx, ry, rl, rw, back, carry, lane, xrow);
}

// out[0..11] = score read_start rmapped genome_start gmapped matches mismatches insertions deletions crossovers n_ops 0;
// ops[] = the reference's backtrace bytes in alignment order (type 1 insertion, 2-5 deletion in layer A-D, 6-9 match/mismatch in layer A-D; | 0x80 crossover)
// (local mode lives in the four-window routine further down: the single call runs it with one group of 16 lanes)
template <int G, typename CT, bool REV, bool TABOO, bool LOCAL>
__device__ CsBest full_sw_cs_g4(const uint8_t* db, int glen, const uint8_t* qr4, int qstride, int rlen, const GmCsDev& P, bool act,
                                int rx, int ry, int rl, int rw, uint32_t* back, CT* carry, int lane, const int8_t* xrow);
__global__ void __launch_bounds__(GM_WAVE)
k_sw_full_cs_single(GmCsDev P, const uint32_t* __restrict__ genome_ls, long long goff, int glen, const uint32_t* __restrict__ read, int rlen, int initbp,
                    int thresh, long long ax, long long ay, int alen, int awidth, int revcmpl, uint32_t* __restrict__ back, int* __restrict__ out,
                    uint8_t* __restrict__ ops, int ops_cap, int local, const int8_t* __restrict__ xrow) {
  extern __shared__ __align__(16) uint8_t sm[];
  const int lane = threadIdx.x;
  const int qstride = (rlen + 15) & ~15;
  uint8_t* rc = sm;                                   // colours
  uint8_t* qr4 = rc + qstride;                        // four translations
  uint8_t* db = qr4 + 4 * qstride;
  int* carry = (int*)(db + ((glen + 15) & ~15));
  load_read(read, rlen, false, rc, lane);
  load_window(genome_ls, (uint64_t)goff, glen, false, db, lane);
  __syncthreads();
  const bool is_rna = (initbp & GM_SEAM_RNA) != 0; initbp &= 0xff;      // (is_rna rides in bit 8 of the primer word)
  if (lane < 4) {                                     // ref :1182-1197
    int letter = (lane + initbp) % 4;
    for (int j = 0; j < rlen; j++) {
      const int base = rc[j];
      if (base == 15) { qr4[lane * qstride + j] = 15; letter = (lane + initbp) % 4; }
      else { const int l2 = cs_cstols(letter, base, is_rna); qr4[lane * qstride + j] = (uint8_t)l2; letter = l2; }
    }
  }
  __syncthreads();
  long long nw = ax + ay, sw = ax - ay, ne = sw + 2 * (awidth - 1), se = nw + 2 * (alen - 1);      // anchor_join + anchor_widen, ref: anchors.c:9-61
  if ((nw + sw) % 2 != 0) nw--;
  long long rx = (nw + sw) / 2, ry = nw - rx;
  if ((ne - sw) % 2 != 0) ne++;
  int rw = (int)((ne - sw) / 2 + 1);
  if ((se - nw) % 2 != 0) se++;
  int rl = (int)((se - nw) / 2 + 1);
  rx -= P.anchor_width / 2; ry += P.anchor_width / 2; rw += P.anchor_width;
  CsBest fo;
  if (local) {                                        // ref: sw-full-cs.c:199-203,315,439-552
    const bool act = lane < 16;
    if (revcmpl) fo = P.taboo > 0 ? full_sw_cs_g4<16, int, true, true, true>(db, glen, qr4, qstride, rlen, P, act, (int)rx, (int)ry, rl, rw, back, carry, lane, xrow)
                                  : full_sw_cs_g4<16, int, true, false, true>(db, glen, qr4, qstride, rlen, P, act, (int)rx, (int)ry, rl, rw, back, carry, lane, xrow);
    else fo = P.taboo > 0 ? full_sw_cs_g4<16, int, false, true, true>(db, glen, qr4, qstride, rlen, P, act, (int)rx, (int)ry, rl, rw, back, carry, lane, xrow)
                          : full_sw_cs_g4<16, int, false, false, true>(db, glen, qr4, qstride, rlen, P, act, (int)rx, (int)ry, rl, rw, back, carry, lane, xrow);
  } else fo = full_sw_cs_wave<true>(db, glen, qr4, qstride, rlen, P, revcmpl != 0, rx, ry, rl, rw, back, carry, lane, xrow);
  __syncthreads();
  __threadfence();
  if (lane == 0) {
    int res[12] = {0};
    if (fo.score >= 0 && fo.score >= thresh) {         // ref :1216; do_backtrace :633-937
      auto code_at = [&](int ci, int cj, int word, int lay) -> int {
        const uint32_t w = __hip_atomic_load(&back[((size_t)ci * glen + cj) * 3 + word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (int)((w >> (8 * lay)) & 0xFFu);
      };
      int i = fo.i, j = fo.j, k = fo.k;
      int from = code_at(i, j, 0, k), fromscore = fo.e_nw;
      if (fo.e_w > fromscore) { from = code_at(i, j, 2, k); fromscore = fo.e_w; }
      if (fo.e_n > fromscore) from = code_at(i, j, 1, k);
      int no = 0, rstart = 0, gstart = 0, nm = 0, nmm = 0, nin = 0, ndel = 0, nx = 0;
      while (i >= 0 && j >= 0 && from != 0) {
        const int dir = from >> 2, lay = from & 3;
        uint8_t bt;
        if (dir == 1 || dir == 2) { ndel++; rstart = i--; bt = (uint8_t)(2 + k); }
        else if (dir == 3 || dir == 4) { nin++; gstart = j--; bt = 1; }
        else {
          const int qv = qr4[k * qstride + i];
          if (db[j] == qv || db[j] == 15 || qv == 15) nm++; else nmm++;
          rstart = i--; gstart = j--; bt = (uint8_t)(6 + k);
        }
        if (k != lay) { bt |= 0x80; nx++; k = lay; }
        if (no < ops_cap) ops[no] = bt;
        no++;
        if (i < 0 || j < 0) break;                    // the virtual row / left sentinel: back == 0 in the reference
        const int word = (dir == 1 || dir == 5) ? 1 : ((dir == 4 || dir == 7) ? 2 : 0);
        from = code_at(i, j, word, k);
      }
      if (k != 0 && no > 0) { if (no - 1 < ops_cap) ops[no - 1] |= 0x80; nx++; }     // ref :929-932
      const int nov = min(no, ops_cap);
      for (int a2 = 0, b2 = nov - 1; a2 < b2; a2++, b2--) { const uint8_t tt = ops[a2]; ops[a2] = ops[b2]; ops[b2] = tt; }
      res[0] = fo.score; res[1] = rstart; res[2] = fo.i - rstart + 1; res[3] = gstart + (int)goff; res[4] = fo.j - gstart + 1;
      res[5] = nm; res[6] = nmm; res[7] = nin; res[8] = ndel; res[9] = nx; res[10] = no;
    }
    for (int x = 0; x < 12; x++) out[x] = res[x];
  }
}

// ---------------------------------------------------------------------------------------------
// K4b in colour space (ref: mapping.c:331-402 with the :375-379 branch): one wave per selected window.  No vector re-score;
// the window is taken on the strand pass 1 left the hit on (strand-1 windows were reversed there, ref :1302-1303), the read is
// translated into its four letter sequences (ref: sw-full-cs.c:1182-1197) and aligned by full_sw_cs_wave; lane 0 walks the
// back pointers (ref :633-937).  Per alignment column the host gets the reference's backtrace byte (ops[0 .. ops_stride/2)) and
// the two 4-bit codes it prints (ops[ops_stride/2 ..): genome letter << 4 | read letter), enough to rebuild dbalign / qralign
// (ref :945-1060) without the genome.
// ---------------------------------------------------------------------------------------------
template <bool TABOO>                               // indel_taboo_len > 0 (a session constant): the taboo tests compiled in
__global__ void __launch_bounds__(GM_WAVE, 2)       // two waves per SIMD: at most 256 registers (the straight-line variants would take 286 and halve the occupancy)
k_pass2_cs(GmIndexDev ix, GmScoreDev sc, GmCsDev P, const uint32_t* __restrict__ reads, const uint8_t* __restrict__ initbp, int n_reads, int read_len,
           int read_words, const GmHit* __restrict__ hits, int hcap, const int32_t* __restrict__ sel,
           const uint32_t* __restrict__ work, const uint32_t* __restrict__ n_work_p, GmFullRes* __restrict__ res, uint8_t* __restrict__ ops, int ops_stride,
           uint32_t* __restrict__ back_pool, size_t back_words, int max_w, unsigned long long* __restrict__ stats,
           const int8_t* __restrict__ xover,           // [n_reads][read_len] crossover scores from the QVs, or null
           const int32_t* __restrict__ sel_sidx) {     // paired mode: the windows' positions in their reads' lists (ref: mapping.c:2545-2552), or null
  extern __shared__ __align__(16) uint8_t sm[];
  const int lane = threadIdx.x;
  const int qstride = (read_len + 15) & ~15;
  uint8_t* rc = sm;                                   // colours of the read
  uint8_t* qr4 = rc + qstride;                        // its four letter translations
  uint8_t* db = qr4 + 4 * qstride;
  int* carry = (int*)(db + ((max_w + 15) & ~15));
  uint32_t* back = back_pool + (size_t)blockIdx.x * back_words;
  const uint32_t n_work = *n_work_p;
  const int half = ops_stride >> 1;
  unsigned long long fcalls = 0, fcells = 0;
  int cur_rd = -1;
  for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
    const uint32_t wk = work[wi];
    const int rd = (int)(wk >> 6), k = (int)(wk & 63);
    const int id = sel[(size_t)rd * SEL_MAX + k];
    const int st = id >> 16, hi = id & 0xFFFF;
    const size_t slot = ((size_t)rd * 2 + st) * hcap + hi;
    const GmHit h = hits[slot];
    __syncthreads();
    if (rd != cur_rd) {
      load_read(reads + (size_t)rd * read_words, read_len, false, rc, lane);
      __syncthreads();
      if (lane < 4) {                                   // ref: sw-full-cs.c:1182-1197
        const int ib = (int)initbp[rd];
        int letter = (lane + ib) % 4;
        for (int j = 0; j < read_len; j++) {
          const int base = rc[j];
          if (base == 15) { qr4[lane * qstride + j] = 15; letter = (lane + ib) % 4; }
          else { const int l2 = cs_cstols(letter, base, ix.genome_is_rna != 0); qr4[lane * qstride + j] = (uint8_t)l2; letter = l2; }
        }
      }
      cur_rd = rd;
    }
    const int cn = h.cn, w_len = h.w_len;
    const long long clen = (long long)ix.contig_off[cn + 1] - ix.contig_off[cn];
    long long g_off = h.g_off; long long ax = h.ax, ay = h.ay; int gen_st = 0;
    if (st != ix.cs_flip) {                             // reverse_hit onto the input strand (label cs_flip), ref: mapping.c:254-263; anchor_reverse anchors.h:30-34
      g_off = clen - g_off - w_len;
      ax = -ax + (w_len - 1) - (h.alen - 1) - (h.awidth - 1);
      ay = -ay + (read_len - 1) - (h.alen - 1) + (h.awidth - 1);
      gen_st = 1;
    }
    load_window(ix.genome, (uint64_t)ix.contig_off[cn] + h.g_off, w_len, gen_st != 0, db, lane, ix.contig_rna && ix.contig_rna[cn]);
    __syncthreads();
    const int score_max = (read_len < w_len ? read_len : w_len) * sc.match;
    const int thresh = thr_of(sc.full_thr_frac, sc.full_abs, score_max);
    GmFullRes R;
    R.read_idx = rd; R.st = (int16_t)ix.cs_flip; R.gen_st = (int16_t)gen_st; R.cn = (uint32_t)cn; R.g_off = (uint32_t)g_off; R.w_len = w_len;
    R.score_vector = h.score_vector; R.score_max = score_max; R.matches = h.matches; R.score_window_gen = h.score_window_gen;
    R.score = 0; R.read_start = 0; R.rmapped = 0; R.genome_start = 0; R.gmapped = 0;
    R.n_match = R.n_mismatch = R.n_ins = R.n_del = 0; R.n_ops = 0; R.ops_off = (uint32_t)(wi * (uint32_t)ops_stride);
    R.sort_idx = sel_sidx ? sel_sidx[(size_t)rd * SEL_MAX + k] : 0; R.hit_slot = (uint32_t)slot; R.n_xover = 0;
    long long nw = ax + ay, sw = ax - ay, ne = sw + 2 * (h.awidth - 1), se = nw + 2 * (h.alen - 1);      // anchor_join + anchor_widen, ref: anchors.c:9-61
    if ((nw + sw) % 2 != 0) nw--;
    long long rx = (nw + sw) / 2, ry = nw - rx;
    if ((ne - sw) % 2 != 0) ne++;
    int rw = (int)((ne - sw) / 2 + 1);
    if ((se - nw) % 2 != 0) se++;
    int rl = (int)((se - nw) / 2 + 1);
    rx -= P.anchor_width / 2; ry += P.anchor_width / 2; rw += P.anchor_width;
    fcalls++; fcells += (unsigned long long)w_len * read_len;
    const CsBest fo = full_sw_cs_wave<TABOO>(db, w_len, qr4, qstride, read_len, P, (gen_st != 0) && sc.tiebreak_rev, rx, ry, rl, rw, back, carry, lane,
                                      xover ? xover + (size_t)rd * read_len : nullptr);
    __syncthreads();
    __threadfence();
    if (lane == 0 && fo.score >= 0 && fo.score >= thresh) {     // ref: sw-full-cs.c:1216; do_backtrace :633-937
      auto code_at = [&](int ci, int cj, int word, int lay) -> int {
        int x_min, x_max; band_range(rx, ry, rl, rw, w_len, ci, &x_min, &x_max);
        if (cj < x_min || cj > x_max) return 0;         // a cell outside the band keeps back == 0 in the reference
        const uint32_t w = __hip_atomic_load(&back[((size_t)ci * w_len + cj) * 3 + word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (int)((w >> (8 * lay)) & 0xFFu);
      };
      uint8_t* o = ops + (size_t)R.ops_off; uint8_t* oc = o + half;
      int i = fo.i, j = fo.j, kk = fo.k;
      int from = code_at(i, j, 0, kk), fromscore = fo.e_nw;
      if (fo.e_w > fromscore) { from = code_at(i, j, 2, kk); fromscore = fo.e_w; }
      if (fo.e_n > fromscore) from = code_at(i, j, 1, kk);
      int no = 0, rstart = 0, gstart = 0, nm = 0, nmm = 0, nin = 0, ndel = 0, nx = 0;
      while (i >= 0 && j >= 0 && from != 0) {
        const int dir = from >> 2, lay = from & 3;
        uint8_t bt, cc;
        if (dir == 1 || dir == 2) { ndel++; cc = qr4[kk * qstride + i]; rstart = i--; bt = (uint8_t)(2 + kk); }
        else if (dir == 3 || dir == 4) { nin++; cc = (uint8_t)(db[j] << 4); gstart = j--; bt = 1; }
        else {
          const int qv = qr4[kk * qstride + i];
          if (db[j] == qv || db[j] == 15 || qv == 15) nm++; else nmm++;
          cc = (uint8_t)((db[j] << 4) | qv);
          rstart = i--; gstart = j--; bt = (uint8_t)(6 + kk);
        }
        if (kk != lay) { bt |= 0x80; nx++; kk = lay; }
        if (no < half) { o[no] = bt; oc[no] = cc; }
        no++;
        if (i < 0 || j < 0) break;                      // the virtual row / left sentinel: back == 0 in the reference
        const int word = (dir == 1 || dir == 5) ? 1 : ((dir == 4 || dir == 7) ? 2 : 0);
        from = code_at(i, j, word, kk);
      }
      if (kk != 0 && no > 0) { if (no - 1 < half) o[no - 1] |= 0x80; nx++; }     // ref :929-932
      const int nov = min(no, half);
      for (int a2 = 0, b2 = nov - 1; a2 < b2; a2++, b2--) { uint8_t tt = o[a2]; o[a2] = o[b2]; o[b2] = tt; tt = oc[a2]; oc[a2] = oc[b2]; oc[b2] = tt; }
      R.score = fo.score; R.n_ops = no; R.read_start = rstart; R.genome_start = gstart + (int)g_off;
      R.gmapped = fo.j - gstart + 1; R.rmapped = fo.i - rstart + 1;
      R.n_match = nm; R.n_mismatch = nmm; R.n_ins = nin; R.n_del = ndel; R.n_xover = nx;
    }
    if (lane == 0) res[wi] = R;
  }
  if (lane == 0) { GS_ADD(stats, GS_FULL_CALLS, fcalls); GS_ADD(stats, GS_FULL_CELLS, fcells); }
}

// ---------------------------------------------------------------------------------------------
// sw_full_cs, four windows per wave (round 3): the scheme of k_pass2_g4 with the colour-space cell -- 16-lane groups, lane = read row 16 s + l, row_shr:1
// between the rows of a group, one kind of window (forward / reverse tie rules) per pass.  Same state updates, back words and traceback as
// full_sw_cs_wave_t / k_pass2_cs (ref: sw-full-cs.c:249-623, :633-937).  The carry values lane 0 of a group needs at the next step are read one step ahead.
// ---------------------------------------------------------------------------------------------
// LOCAL (Gflag off, ref: sw-full-cs.c:199-203,315,439-552): a cell outside the band holds (0, -b_open, -a_open) -- plus the crossover score in layers 1-3 -- like the
// virtual row above the matrix; a state at or below 0 (layer 0) / the crossover score (layers 1-3) takes that value with a null back pointer; the result is the
// first cell in row-major order with the largest score.  (With per-position crossover scores an out-of-band cell carries the score of its own row, as the reference's init_cell leaves it: sw-full-cs.c:312-322.)
#ifdef P2CS_STAMPS
__device__ unsigned long long p2cs_stamps[8];
#endif
template <int G, typename CT, bool REV, bool TABOO, bool LOCAL>
__device__ CsBest full_sw_cs_g4(const uint8_t* db, int glen, const uint8_t* qr4, int qstride, int rlen, const GmCsDev& P, bool act,
                                int rx, int ry, int rl, int rw, uint32_t* back, CT* carry, int lane, const int8_t* xrow) {
  constexpr bool revcmpl = REV;
  // a cell outside the band: what the reference's init_cell leaves there -- in local mode with the crossover score of the cell's ROW (per-position scores from quality values, ref: sw-full-cs.c:312-322)
  auto OBx = [&](const int x, const int xv) -> int { const int k = x / 3, st = x % 3; return LOCAL ? (st == 0 ? 0 : (st == 1 ? -P.b_go : -P.a_go)) + (k ? xv : 0) : FS_NEG; };
  CsBest best; best.score = 0; best.i = best.j = best.k = 0; best.e_nw = best.e_n = best.e_w = 0;
  const int l = lane & (G - 1);
  const int xg = P.xover;
  int xo = xg;
  const int n_stripes = (rlen + G - 1) / G;
  int cw_lo = 1, cw_hi = 0;
  for (int s = 0; s < n_stripes; s++) {
    const int r = s * G + l;
    const bool row_ok = act && r < rlen;
    if (xrow) xo = row_ok ? (int)xrow[r] : xg;           // ref: sw-full-cs.c:312
    const int xo_up = (xrow && act && r >= 1 && r - 1 < rlen) ? (int)xrow[r - 1] : xg;      // the row above's
    auto OB = [&](const int x) -> int { return OBx(x, xo); };
    int q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = row_ok ? qr4[k * qstride + r] : 0x7F;
    int x_min = 0, x_max = -1;
    if (row_ok) band_range(rx, ry, rl, rw, glen, r, &x_min, &x_max);
    const bool notaboo = TABOO ? r < rlen - P.taboo : true;
    int t_lo = INT_MAX, t_hi = -1;
    if (row_ok && x_max >= x_min) { t_lo = x_min + l; t_hi = x_max + l; }
    for (int dd = G / 2; dd > 0; dd >>= 1) { t_lo = min(t_lo, __shfl_xor(t_lo, dd)); t_hi = max(t_hi, __shfl_xor(t_hi, dd)); }
    int nst = t_hi >= 0 ? t_hi - t_lo + 1 : 0, nmax = nst;
    for (int dd = 32; dd >= G; dd >>= 1) nmax = max(nmax, __shfl_xor(nmax, dd));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
#ifdef P2CS_STAMPS
    if (lane == 0) atomicAdd(&p2cs_stamps[4], (unsigned long long)nmax);
    if (l == 0) atomicAdd(&p2cs_stamps[5], (unsigned long long)nst);
    if (row_ok && x_max >= x_min) atomicAdd(&p2cs_stamps[6], (unsigned long long)(x_max - x_min + 1));
#endif
    int d[12], cur[12], inv[12];        // (cur: this lane's cell of the step before = the west neighbour of the next one)
#pragma unroll
    for (int x = 0; x < 12; x++) { d[x] = OBx(x, xo_up); cur[x] = OB(x); }
    // lane 0's upper neighbour: the virtual row (stripe 0: constants), or the previous stripe's last row (read one step ahead below)
#pragma unroll
    for (int x = 0; x < 12; x++) { const int k = x / 3, st = x % 3; const int xv = k ? xg : 0; inv[x] = s == 0 ? (st == 0 ? 0 : (st == 1 ? -P.b_go : -P.a_go)) + xv : OBx(x, xo_up); }
    if (l == 0) {
      if (s == 0) {                                  // virtual row -1: init_cell(.., 1, xover), ref :201-215
#pragma unroll
        for (int k = 0; k < 4; k++) { const int x = k ? xg : 0; d[k * 3] = x; d[k * 3 + 1] = -P.b_go + x; d[k * 3 + 2] = -P.a_go + x; }
      } else if (t_hi >= 0) {
        if (t_lo >= 1 && t_lo - 1 >= cw_lo && t_lo - 1 <= cw_hi) {
#pragma unroll
          for (int x = 0; x < 12; x++) d[x] = cs_carry_ld(carry[x * glen + t_lo - 1]);
        }
        if (t_lo >= cw_lo && t_lo <= cw_hi) {
#pragma unroll
          for (int x = 0; x < 12; x++) inv[x] = cs_carry_ld(carry[x * glen + t_lo]);
        }
      }
    }
    const bool more = (s + 1 < n_stripes);
    const bool last_row_lane = row_ok && (r == rlen - 1);
    // One step of the stripe.  d / cur: the cells (r-1, c-1) and (r, c-1) of this lane; the step leaves (r-1, c) and (r, c) in u / nv, which are the next step's d / cur: the
    // loop below runs two steps per round with the two register sets swapped, so that no values are copied from one set to the other (24 of a step's ~360 instructions).
    auto step = [&](const int i, const int (&d)[12], const int (&cur)[12], int (&u)[12], int (&nv)[12]) {
      const bool on = i < nst;
      const int t = t_lo + i;
      const int c = t - l;
#pragma unroll
      for (int x = 0; x < 12; x++) u[x] = gn_shr1<G>(cur[x], inv[x], l);   // cell (r-1, c)
      if (s > 0) {                                   // next step's carry values for the group's first lane
#pragma unroll
        for (int x = 0; x < 12; x++) inv[x] = OBx(x, xo_up);
        if (l == 0 && i + 1 < nst && t + 1 >= cw_lo && t + 1 <= cw_hi) {
#pragma unroll
          for (int x = 0; x < 12; x++) inv[x] = cs_carry_ld(carry[x * glen + t + 1]);
        }
      }
      const bool inband = on && row_ok && c >= x_min && c <= x_max;
      // (every lane works the cell out, the band decides afterwards what is kept: straight-line code -- a branch around the cell cost more register copies at its join
      // than the cell has comparisons)
      {
        const int dbc = db[min(max(c, 0), glen - 1)];
        uint32_t bw_nw = 0, bw_n = 0, bw_w = 0;
        // Without an indel taboo the reference's scans factor: every candidate of the diagonal move is "a state of some layer (+ a crossover when the layer changes)", tried in
        // a fixed order with strict comparisons (the first maximum wins), and the own layer comes first -- so the first maximum over the whole list is the first maximum over
        // the layers' own first maxima.  The per-layer maxima (value and back code) are the same for all four target layers: 8 + 4 compare-selects per cell for them, 3 per
        // target layer for the choice among layers, instead of 11 (diagonal) and 7 (vertical) per target layer.
        int Mv[4], Mc[4], Nv[4], Nc[4];
        if (!TABOO) {
#pragma unroll
          for (int l2 = 0; l2 < 4; l2++) {
            int v, c;
            if (!revcmpl) { v = d[l2 * 3]; c = 6; if (d[l2 * 3 + 1] > v) { v = d[l2 * 3 + 1]; c = 5; } if (d[l2 * 3 + 2] > v) { v = d[l2 * 3 + 2]; c = 7; } }
            else { v = d[l2 * 3 + 2]; c = 7; if (d[l2 * 3 + 1] > v) { v = d[l2 * 3 + 1]; c = 5; } if (d[l2 * 3] > v) { v = d[l2 * 3]; c = 6; } }
            Mv[l2] = v; Mc[l2] = (c << 2) | l2;
            const int op = u[l2 * 3] - P.b_go - P.b_ge, ex = u[l2 * 3 + 1] - P.b_ge;
            if (!revcmpl) { v = op; c = 2; if (ex > v) { v = ex; c = 1; } } else { v = ex; c = 1; if (op > v) { v = op; c = 2; } }
            Nv[l2] = v; Nc[l2] = (c << 2) | l2;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int ms = (dbc == 15 || q[k] == 15) ? 0 : (dbc == q[k] ? P.match : P.mismatch);
          int tmp, b;
          // northwest, ref :356-438
          if (!TABOO) {
            tmp = Mv[k]; b = Mc[k];
#pragma unroll
            for (int l2 = 0; l2 < 4; l2++) { if (l2 == k) continue; if (Mv[l2] + xo > tmp) { tmp = Mv[l2] + xo; b = Mc[l2]; } }
            tmp += ms;
          } else
          {
          if (!revcmpl) {
            tmp = d[k * 3] + ms; b = (6 << 2) | k;
            if (notaboo && d[k * 3 + 1] + ms > tmp) { tmp = d[k * 3 + 1] + ms; b = (5 << 2) | k; }
            if (d[k * 3 + 2] + ms > tmp) { tmp = d[k * 3 + 2] + ms; b = (7 << 2) | k; }
          } else {
            tmp = d[k * 3 + 2] + ms; b = (7 << 2) | k;
            if (notaboo && d[k * 3 + 1] + ms > tmp) { tmp = d[k * 3 + 1] + ms; b = (5 << 2) | k; }
            if (d[k * 3] + ms > tmp) { tmp = d[k * 3] + ms; b = (6 << 2) | k; }
          }
#pragma unroll
          for (int l2 = 0; l2 < 4; l2++) {
            if (l2 == k) continue;
            if (!revcmpl) {
              if (d[l2 * 3] + ms + xo > tmp) { tmp = d[l2 * 3] + ms + xo; b = (6 << 2) | l2; }
              if (notaboo && d[l2 * 3 + 1] + ms + xo > tmp) { tmp = d[l2 * 3 + 1] + ms + xo; b = (5 << 2) | l2; }
              if (d[l2 * 3 + 2] + ms + xo > tmp) { tmp = d[l2 * 3 + 2] + ms + xo; b = (7 << 2) | l2; }
            } else {
              if (d[l2 * 3 + 2] + ms + xo > tmp) { tmp = d[l2 * 3 + 2] + ms + xo; b = (7 << 2) | l2; }
              if (notaboo && d[l2 * 3 + 1] + ms + xo > tmp) { tmp = d[l2 * 3 + 1] + ms + xo; b = (5 << 2) | l2; }
              if (d[l2 * 3] + ms + xo > tmp) { tmp = d[l2 * 3] + ms + xo; b = (6 << 2) | l2; }
            }
          }
          }
          const int resetval = k ? xo : 0;               // :350-353
          if (LOCAL && tmp <= resetval) { tmp = resetval; b = 0; }
          nv[k * 3] = tmp; bw_nw |= (uint32_t)b << (8 * k);
          // north, ref :447-503
          if (!TABOO) {
            tmp = Nv[k]; b = Nc[k];
#pragma unroll
            for (int l2 = 0; l2 < 4; l2++) { if (l2 == k) continue; if (Nv[l2] + xo > tmp) { tmp = Nv[l2] + xo; b = Nc[l2]; } }
          } else
          {
          if (!revcmpl) {
            tmp = u[k * 3] - P.b_go - P.b_ge; b = (2 << 2) | k;
            if (!notaboo || u[k * 3 + 1] - P.b_ge > tmp) { tmp = u[k * 3 + 1] - P.b_ge; b = (1 << 2) | k; }
          } else {
            tmp = u[k * 3 + 1] - P.b_ge; b = (1 << 2) | k;
            if (notaboo && u[k * 3] - P.b_go - P.b_ge > tmp) { tmp = u[k * 3] - P.b_go - P.b_ge; b = (2 << 2) | k; }
          }
#pragma unroll
          for (int l2 = 0; l2 < 4; l2++) {
            if (l2 == k) continue;
            if (!revcmpl) {
              if (notaboo && u[l2 * 3] - P.b_go - P.b_ge + xo > tmp) { tmp = u[l2 * 3] - P.b_go - P.b_ge + xo; b = (2 << 2) | l2; }
              if (u[l2 * 3 + 1] - P.b_ge + xo > tmp) { tmp = u[l2 * 3 + 1] - P.b_ge + xo; b = (1 << 2) | l2; }
            } else {
              if (u[l2 * 3 + 1] - P.b_ge + xo > tmp) { tmp = u[l2 * 3 + 1] - P.b_ge + xo; b = (1 << 2) | l2; }
              if (notaboo && u[l2 * 3] - P.b_go - P.b_ge + xo > tmp) { tmp = u[l2 * 3] - P.b_go - P.b_ge + xo; b = (2 << 2) | l2; }
            }
          }
          }
          if (LOCAL && tmp <= resetval) { tmp = resetval; b = 0; }
          nv[k * 3 + 1] = tmp; bw_n |= (uint32_t)b << (8 * k);
          // west, ref :512-541 (no crossover on a genomic gap)
          if (!revcmpl) {
            tmp = cur[k * 3] - P.a_go - P.a_ge; b = (3 << 2) | k;
            if (!notaboo || cur[k * 3 + 2] - P.a_ge > tmp) { tmp = cur[k * 3 + 2] - P.a_ge; b = (4 << 2) | k; }
          } else {
            tmp = cur[k * 3 + 2] - P.a_ge; b = (4 << 2) | k;
            if (notaboo && cur[k * 3] - P.a_go - P.a_ge > tmp) { tmp = cur[k * 3] - P.a_go - P.a_ge; b = (3 << 2) | k; }
          }
          if (LOCAL && tmp <= resetval) { tmp = resetval; b = 0; }
          nv[k * 3 + 2] = tmp; bw_w |= (uint32_t)b << (8 * k);
          if (inband && (LOCAL || last_row_lane)) {    // ref :547-575 (local: every row)
            const int a0 = revcmpl ? nv[k * 3 + 2] : nv[k * 3], a1 = nv[k * 3 + 1], a2 = revcmpl ? nv[k * 3] : nv[k * 3 + 2];
            const int m = max(a0, max(a1, a2));
            if (m > best.score) { best.score = m; best.i = r; best.j = c; best.k = k; best.e_nw = nv[k * 3]; best.e_n = nv[k * 3 + 1]; best.e_w = nv[k * 3 + 2]; }
          }
        }
        if (inband) {
          uint32_t* bp = back + ((size_t)r * glen + c) * 3;
          bp[0] = bw_nw; bp[1] = bw_n; bp[2] = bw_w;
        }
      }
#pragma unroll
      for (int x = 0; x < 12; x++) nv[x] = inband ? nv[x] : OB(x);
      if (more && l == G - 1 && on && c >= 0 && c < glen) {
#pragma unroll
        for (int x = 0; x < 12; x++) cs_carry_st(carry[x * glen + c], nv[x]);
      }
    };
    {
      int d2[12], cur2[12];
      int i = 0;
      for (; i + 1 < nmax; i += 2) { step(i, d, cur, d2, cur2); step(i + 1, d2, cur2, d, cur); }
      if (i < nmax) step(i, d, cur, d2, cur2);
    }
    cw_lo = 1; cw_hi = 0;
    if (more && t_hi >= 0) { cw_lo = max(0, t_lo - (G - 1)); cw_hi = min(glen - 1, t_hi - (G - 1)); }
    if (more) __syncthreads();
  }
  int src = (lane & (64 - G)) | ((rlen - 1) & (G - 1));
  if (LOCAL) {                                         // the lane whose row comes first among those with the largest score
    int bs = best.score;
    for (int dd = G / 2; dd > 0; dd >>= 1) bs = max(bs, __shfl_xor(bs, dd));
    int row = (best.score == bs) ? best.i : INT_MAX;
    for (int dd = G / 2; dd > 0; dd >>= 1) row = min(row, __shfl_xor(row, dd));
    src = (lane & (64 - G)) | ((bs > 0) ? (row & (G - 1)) : 0);
  }
  best.score = __shfl(best.score, src); best.i = __shfl(best.i, src); best.j = __shfl(best.j, src); best.k = __shfl(best.k, src);
  best.e_nw = __shfl(best.e_nw, src); best.e_n = __shfl(best.e_n, src); best.e_w = __shfl(best.e_w, src);
  return best;
}

// Diagnostic build (-DP2CS_STAMPS, tools/p2cs_stamps.py): lane 0 of every wave adds the cycles of a pass's phases to p2cs_stamps[] (0: set-up -- unpack read and window,
// translations; 1: the cells; 2: traceback; 3: passes).  No stamp executes in the normal build.
#ifdef P2CS_STAMPS
#define P2CS_STAMP(i) do { if (lane == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&p2cs_stamps[i], t_ - t_prev); t_prev = t_; } } while (0)
extern "C" int gm_debug_p2cs_stamps(unsigned long long* out) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(p2cs_stamps), sizeof z) != hipSuccess) return GM_E_NODEVICE;
  if (hipMemcpyToSymbol(HIP_SYMBOL(p2cs_stamps), z, sizeof z) != hipSuccess) return GM_E_NODEVICE;
  return GM_OK;
}
#else
#define P2CS_STAMP(i) do { } while (0)
#endif
struct P2CsG4 {
  const uint32_t* reads; const uint8_t* initbp; int read_len, read_words; const GmHit* hits; int hcap; const int32_t* sel; const int32_t* sel_sidx;
  const uint32_t* work; GmFullRes* res; uint8_t* ops; int ops_stride, max_w; const int8_t* xover;
  uint8_t* rc_all; uint8_t* qr4_all; uint8_t* db_all; void* carry_all; int qstride, mw16;
};
template <int G, typename CT, bool REV, bool TABOO, bool LOCAL>
__device__ __forceinline__ void p2cs_g4_pass(const GmIndexDev& ix, const GmScoreDev& sc, const GmCsDev& P, const P2CsG4& A, const uint32_t wi, const bool has, uint32_t* back,
                                             const int lane, unsigned long long& fcalls, unsigned long long& fcells) {
  constexpr int NG = 64 / G;
  const int g = lane / G, l = lane & (G - 1);
#ifdef P2CS_STAMPS
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  const int read_len = A.read_len, qstride = A.qstride, half = A.ops_stride >> 1;
  const uint8_t* qr4 = A.qr4_all + g * 4 * qstride; const uint8_t* db = A.db_all + g * A.mw16; CT* carry = (CT*)A.carry_all + g * 12 * A.max_w;
  const uint32_t wk = has ? A.work[wi] : 0u;
  const int rd = (int)(wk >> 6), k = (int)(wk & 63);
  const int id = has ? A.sel[(size_t)rd * SEL_MAX + k] : 0;
  const int st = id >> 16, hi = id & 0xFFFF;
  const size_t slot = ((size_t)rd * 2 + st) * A.hcap + hi;
  GmHit h; if (has) h = A.hits[slot]; else { h.g_off = 0; h.ax = h.ay = 0; h.alen = h.awidth = 1; h.score_window_gen = 0; h.score_vector = 0; h.pct_score_vector = 0; h.cn = 0; h.w_len = 1; h.matches = 0; h.flags = 1; }
  const int cn = h.cn, w_len = h.w_len;
  const long long clen = (long long)ix.contig_off[cn + 1] - ix.contig_off[cn];
  long long g_off = h.g_off; long long ax = h.ax, ay = h.ay; int gen_st = 0;
  if (st != ix.cs_flip) {                               // reverse_hit onto the input strand (label cs_flip), ref: mapping.c:254-263; anchor_reverse anchors.h:30-34
    g_off = clen - g_off - w_len;
    ax = -ax + (w_len - 1) - (h.alen - 1) - (h.awidth - 1);
    ay = -ay + (read_len - 1) - (h.alen - 1) + (h.awidth - 1);
    gen_st = 1;
  }
  const uint64_t g0 = (uint64_t)ix.contig_off[cn] + h.g_off;
  const int rna_bits = (ix.contig_rna && has && ix.contig_rna[cn]) ? 2 : 0;
  __syncthreads();
  for (int gg = 0; gg < NG; gg++) {                      // the whole wave unpacks each group's colours and window
    if (!__shfl((int)has, gg * G)) continue;
    const int rd_g = __shfl(rd, gg * G), wl_g = __shfl(w_len, gg * G), gs_g = __builtin_amdgcn_readfirstlane(__shfl(gen_st | rna_bits, gg * G));      // (bit 1: the contig is RNA; wave-uniform)
    const uint64_t g0_g = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(g0 >> 32), gg * G) << 32) | (uint32_t)__shfl((int)(uint32_t)g0, gg * G);
    load_read(A.reads + (size_t)rd_g * A.read_words, read_len, false, A.rc_all + gg * qstride, lane);
    load_window(ix.genome, g0_g, wl_g, (gs_g & 1) != 0, A.db_all + gg * A.mw16, lane, (gs_g & 2) != 0);
  }
  __syncthreads();
  if (has && l < 4) {                                    // the four letter translations of the group's read, ref: sw-full-cs.c:1182-1197
    const uint8_t* rc = A.rc_all + g * qstride; uint8_t* q4 = A.qr4_all + g * 4 * qstride;
    const int ib = (int)A.initbp[rd];
    int letter = (l + ib) % 4;
    for (int j = 0; j < read_len; j++) {
      const int base = rc[j];
      if (base == 15) { q4[l * qstride + j] = 15; letter = (l + ib) % 4; }
      else { const int l2 = cs_cstols(letter, base, ix.genome_is_rna != 0); q4[l * qstride + j] = (uint8_t)l2; letter = l2; }
    }
  }
  const int score_max = (read_len < w_len ? read_len : w_len) * sc.match;
  const int thresh = thr_of(sc.full_thr_frac, sc.full_abs, score_max);
  const uint32_t ops_off = (uint32_t)(wi * (uint32_t)A.ops_stride);
  if (has && l == 0) {
    GmFullRes R;
    R.read_idx = rd; R.st = (int16_t)ix.cs_flip; R.gen_st = (int16_t)gen_st; R.cn = (uint32_t)cn; R.g_off = (uint32_t)g_off; R.w_len = w_len;
    R.score_vector = h.score_vector; R.score_max = score_max; R.matches = h.matches; R.score_window_gen = h.score_window_gen;
    R.score = 0; R.read_start = 0; R.rmapped = 0; R.genome_start = 0; R.gmapped = 0;
    R.n_match = R.n_mismatch = R.n_ins = R.n_del = 0; R.n_ops = 0; R.ops_off = ops_off;
    R.sort_idx = A.sel_sidx ? A.sel_sidx[(size_t)rd * SEL_MAX + k] : 0; R.hit_slot = (uint32_t)slot; R.n_xover = 0;
    A.res[wi] = R;
    fcalls++; fcells += (unsigned long long)w_len * read_len;
  }
  int rx, ry, rw, rl;
  { long long nw = ax + ay, sw = ax - ay, ne = sw + 2 * (h.awidth - 1), se = nw + 2 * (h.alen - 1);      // anchor_join + anchor_widen, ref: anchors.c:9-61
    if ((nw + sw) % 2 != 0) nw--;
    const long long x0 = (nw + sw) / 2;
    rx = (int)x0; ry = (int)(nw - x0);
    if ((ne - sw) % 2 != 0) ne++;
    rw = (int)((ne - sw) / 2 + 1);
    if ((se - nw) % 2 != 0) se++;
    rl = (int)((se - nw) / 2 + 1);
    rx -= P.anchor_width / 2; ry += P.anchor_width / 2; rw += P.anchor_width; }
  const int g_off_i = (int)g_off;
  __syncthreads();
  P2CS_STAMP(0);
  const CsBest fo = full_sw_cs_g4<G, CT, REV, TABOO, LOCAL>(db, w_len, qr4, qstride, read_len, P, has, rx, ry, rl, rw, back, carry, lane, A.xover ? A.xover + (size_t)rd * read_len : nullptr);
  __syncthreads();
  __threadfence();
  P2CS_STAMP(1);
  if (has && l == 0 && fo.score >= 0 && fo.score >= thresh) {     // ref: sw-full-cs.c:1216; do_backtrace :633-937
    auto code_at = [&](int ci, int cj, int word, int lay) -> int {
      int x_min, x_max; band_range(rx, ry, rl, rw, w_len, ci, &x_min, &x_max);
      if (cj < x_min || cj > x_max) return 0;         // a cell outside the band keeps back == 0 in the reference
      const uint32_t w = __hip_atomic_load(&back[((size_t)ci * w_len + cj) * 3 + word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return (int)((w >> (8 * lay)) & 0xFFu);
    };
    uint8_t* o = A.ops + (size_t)ops_off; uint8_t* oc = o + half;
    int i = fo.i, j = fo.j, kk = fo.k;
    int from = code_at(i, j, 0, kk), fromscore = fo.e_nw;
    if (fo.e_w > fromscore) { from = code_at(i, j, 2, kk); fromscore = fo.e_w; }
    if (fo.e_n > fromscore) from = code_at(i, j, 1, kk);
    int no = 0, rstart = 0, gstart = 0, nm = 0, nmm = 0, nin = 0, ndel = 0, nx = 0;
    while (i >= 0 && j >= 0 && from != 0) {
      const int dir = from >> 2, lay = from & 3;
      uint8_t bt, cc;
      if (dir == 1 || dir == 2) { ndel++; cc = qr4[kk * qstride + i]; rstart = i--; bt = (uint8_t)(2 + kk); }
      else if (dir == 3 || dir == 4) { nin++; cc = (uint8_t)(db[j] << 4); gstart = j--; bt = 1; }
      else {
        const int qv = qr4[kk * qstride + i];
        if (db[j] == qv || db[j] == 15 || qv == 15) nm++; else nmm++;
        cc = (uint8_t)((db[j] << 4) | qv);
        rstart = i--; gstart = j--; bt = (uint8_t)(6 + kk);
      }
      if (kk != lay) { bt |= 0x80; nx++; kk = lay; }
      if (no < half) { o[no] = bt; oc[no] = cc; }
      no++;
      if (i < 0 || j < 0) break;                      // the virtual row / left sentinel: back == 0 in the reference
      const int word = (dir == 1 || dir == 5) ? 1 : ((dir == 4 || dir == 7) ? 2 : 0);
      from = code_at(i, j, word, kk);
    }
    if (kk != 0 && no > 0) { if (no - 1 < half) o[no - 1] |= 0x80; nx++; }     // ref :929-932
    const int nov = min(no, half);
    for (int a2 = 0, b2 = nov - 1; a2 < b2; a2++, b2--) { uint8_t tt = o[a2]; o[a2] = o[b2]; o[b2] = tt; tt = oc[a2]; oc[a2] = oc[b2]; oc[b2] = tt; }
    GmFullRes* Rp = &A.res[wi];
    Rp->score = fo.score; Rp->n_ops = no; Rp->read_start = rstart; Rp->genome_start = gstart + g_off_i;
    Rp->gmapped = fo.j - gstart + 1; Rp->rmapped = fo.i - rstart + 1;
    Rp->n_match = nm; Rp->n_mismatch = nmm; Rp->n_ins = nin; Rp->n_del = ndel; Rp->n_xover = nx;
  }
  __builtin_amdgcn_wave_barrier();
  P2CS_STAMP(2);
#ifdef P2CS_STAMPS
  if (lane == 0) atomicAdd(&p2cs_stamps[3], 1ull);
#endif
}

template <int G, typename CT, bool TABOO, bool LOCAL>
__global__ void __launch_bounds__(GM_WAVE, 2)
k_pass2_cs_g4(GmIndexDev ix, GmScoreDev sc, GmCsDev P, const uint32_t* __restrict__ reads, const uint8_t* __restrict__ initbp, int n_reads, int read_len,
              int read_words, const GmHit* __restrict__ hits, int hcap, const int32_t* __restrict__ sel,
              const uint32_t* __restrict__ work, const uint32_t* __restrict__ n_work_p, GmFullRes* __restrict__ res, uint8_t* __restrict__ ops, int ops_stride,
              uint32_t* __restrict__ back_pool, size_t back_words, int max_w, unsigned long long* __restrict__ stats,
              const int8_t* __restrict__ xover, const int32_t* __restrict__ sel_sidx, const uint32_t* __restrict__ order, uint32_t* __restrict__ cls_cnt) {
  extern __shared__ __align__(16) uint8_t sm[];
  constexpr int NG = 64 / G;
  const int lane = threadIdx.x, g = lane / G;
  P2CsG4 A;
  A.reads = reads; A.initbp = initbp; A.read_len = read_len; A.read_words = read_words; A.hits = hits; A.hcap = hcap; A.sel = sel; A.sel_sidx = sel_sidx;
  A.work = work; A.res = res; A.ops = ops; A.ops_stride = ops_stride; A.max_w = max_w; A.xover = xover;
  A.qstride = (read_len + 15) & ~15; A.mw16 = (max_w + 15) & ~15;
  A.rc_all = sm; A.qr4_all = sm + NG * A.qstride; A.db_all = A.qr4_all + 4 * NG * A.qstride; A.carry_all = (void*)(A.db_all + NG * A.mw16);
  uint32_t* back = back_pool + ((size_t)blockIdx.x * NG + g) * back_words;
  const uint32_t n_work = *n_work_p;
  unsigned long long fcalls = 0, fcells = 0;
  // A pass holds windows of one kind (the strand decides the tie rules of all 36 state updates of a cell, ref: sw-full-cs.c:356-541; as a per-lane value both variants of every
  // update would run).  k_p2cs_classify has listed the work items by kind -- forward ones from the front of order[], reverse ones from its back -- so a pass is one unit of NG
  // list entries of one kind, full except for the last of each kind, and the waves draw units from a counter until none is left (round 4: chunks of 32 consecutive items split by
  // kind filled 82 % of eight window slots, and 4 486 such chunks over 2 048 resident waves left a third of the waves idle in the last round).
  const uint32_t n_fw = cls_cnt[0], n_rv = cls_cnt[1];
  const uint32_t u_fw = (n_fw + NG - 1) / NG, u_all = u_fw + (n_rv + NG - 1) / NG;
  for (;;) {
    uint32_t u = 0;
    if (lane == 0) u = atomicAdd(&cls_cnt[2], 1u);
    u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
    if (u >= u_all) break;
    if (u < u_fw) {
      const uint32_t idx = u * NG + (uint32_t)g;
      const bool has = idx < n_fw;
      const uint32_t wi = has ? order[idx] : 0u;
      p2cs_g4_pass<G, CT, false, TABOO, LOCAL>(ix, sc, P, A, wi, has, back, lane, fcalls, fcells);
    } else {
      const uint32_t idx = (u - u_fw) * NG + (uint32_t)g;
      const bool has = idx < n_rv;
      const uint32_t wi = has ? order[n_work - 1u - idx] : 0u;
      p2cs_g4_pass<G, CT, true, TABOO, LOCAL>(ix, sc, P, A, wi, has, back, lane, fcalls, fcells);
    }
  }
  for (int d = 32; d >= G; d >>= 1) { fcalls += __shfl_xor(fcalls, d); fcells += __shfl_xor(fcells, d); }
  if (lane == 0) { GS_ADD(stats, GS_FULL_CALLS, fcalls); GS_ADD(stats, GS_FULL_CELLS, fcells); }
}

int gm_launch_pass2_cs(const GmIndexDev& ix, const GmScoreDev& sc, const int* cs_params9, const uint32_t* d_reads, const uint8_t* d_initbp, int n_reads,
                       int read_len, int read_words, int window_len, const GmHit* d_hits, int hcap, const int32_t* d_sel, const uint32_t* d_work,
                       const uint32_t* d_n_work, GmFullRes* d_res, uint8_t* d_ops, int ops_stride, uint32_t* d_back, size_t back_words, int grid,
                       unsigned long long* d_stats, hipStream_t stream, const int8_t* d_xover, const int32_t* d_sel_sidx, uint32_t* d_order, uint32_t* d_cls_cnt) {
  if (n_reads == 0) return GM_OK;
  GmCsDev P; P.match = cs_params9[0]; P.mismatch = cs_params9[1]; P.xover = cs_params9[2]; P.a_go = cs_params9[3]; P.a_ge = cs_params9[4];
  P.b_go = cs_params9[5]; P.b_ge = cs_params9[6]; P.anchor_width = cs_params9[7]; P.taboo = cs_params9[8];
  const size_t lds = 5 * (size_t)((read_len + 15) & ~15) + ((window_len + 15) & ~15) + (size_t)window_len * 48 + 64;
  if (lds > 160 * 1024) { gm_set_error("colour-space pass 2: window of %d does not fit LDS", window_len); return GM_E_ARG; }
  static GmLdsLimit lim_configured; size_t& configured = lim_configured.cur();
  if (lds > 48 * 1024 && lds > configured) {
    GM_HIP(hipFuncSetAttribute((const void*)k_pass2_cs<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GM_HIP(hipFuncSetAttribute((const void*)k_pass2_cs<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); configured = lds; }
  // Four or eight windows per wave (k_pass2_cs_g4) unless GM_P2_G4=0 asks for the one-window kernel; a wave owns that many consecutive back-pointer scratches.
  // Eight (groups of 8 lanes: a stripe of 8 rows takes band width + 14 steps, against band width + 30 for 16 rows -- the chain north, west, north, ... through a band is two steps a
  // row whatever the lane count, so ~band width / 2 lanes a window is what the recurrence can feed) where the carry rows fit int16_t, see cs_carry_ld; GM_P2_G=16 keeps four.
  // Local alignment (sc.local, ref: sw-full-cs.c:199-203,439-552) exists in these kernels only.
  const bool want_g4 = !(gm_tune("GM_P2_G4") && atoi(gm_tune("GM_P2_G4")) == 0) && grid >= 8;
  if (want_g4 || sc.local) {
    const size_t q16 = (size_t)((read_len + 15) & ~15), w16 = (size_t)((window_len + 15) & ~15);
    // no score on a path through the matrix lies further from 0 than `reach`: a path has at most read_len + window_len moves, and a move changes the score by a match or a
    // mismatch plus a crossover (per-position crossover scores lie in [2 x crossover_score, -1], ref: gmapper.c:532-544; an int8 row), or by a gap's opening + extension, at most
    const int xmax = d_xover ? std::min(127, 2 * std::abs(P.xover)) : std::abs(P.xover);
    const int big = std::max(std::max(std::abs(P.match), std::abs(P.mismatch)) + xmax, std::max(std::max(std::abs(P.a_go) + std::abs(P.a_ge), std::abs(P.b_go) + std::abs(P.b_ge)), 1));
    const long long reach = (long long)(read_len + window_len) * big;
    const size_t lds8 = 8 * 5 * q16 + 8 * w16 + 8 * (size_t)window_len * 12 * sizeof(int16_t) + 64;
    const size_t lds4 = 4 * 5 * q16 + 4 * w16 + 4 * (size_t)window_len * 12 * sizeof(int) + 64;
    const bool g8 = reach < 16000 && lds8 <= 64 * 1024 && !(gm_tune("GM_P2_G") && atoi(gm_tune("GM_P2_G")) == 16);
    const size_t ldsn = g8 ? lds8 : lds4;
    if (ldsn <= 160 * 1024 && grid >= 8) {
      static GmLdsLimit lim4; size_t& conf4 = lim4.cur();
      if (ldsn > 48 * 1024 && ldsn > conf4) {
        GM_HIP(hipFuncSetAttribute((const void*)k_pass2_cs_g4<16, int, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
        GM_HIP(hipFuncSetAttribute((const void*)k_pass2_cs_g4<16, int, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
        GM_HIP(hipFuncSetAttribute((const void*)k_pass2_cs_g4<16, int, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
        GM_HIP(hipFuncSetAttribute((const void*)k_pass2_cs_g4<16, int, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
        GM_HIP(hipFuncSetAttribute((const void*)k_pass2_cs_g4<8, int16_t, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
        GM_HIP(hipFuncSetAttribute((const void*)k_pass2_cs_g4<8, int16_t, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
        GM_HIP(hipFuncSetAttribute((const void*)k_pass2_cs_g4<8, int16_t, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
        GM_HIP(hipFuncSetAttribute((const void*)k_pass2_cs_g4<8, int16_t, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn)); conf4 = ldsn; }
#define GM_P2CS_G4(GG, CT, TB, LOC) hipLaunchKernelGGL((k_pass2_cs_g4<GG, CT, TB, LOC>), dim3(grid / (64 / GG)), dim3(GM_WAVE), ldsn, stream, ix, sc, P, d_reads, d_initbp, n_reads, read_len, read_words, d_hits, hcap, d_sel, \
                           d_work, d_n_work, d_res, d_ops, ops_stride, d_back, back_words, window_len, d_stats, d_xover, d_sel_sidx, d_order, d_cls_cnt)
#define GM_P2CS_GN(TB, LOC) do { if (g8) GM_P2CS_G4(8, int16_t, TB, LOC); else GM_P2CS_G4(16, int, TB, LOC); } while (0)
      if (!d_order || !d_cls_cnt) { gm_set_error("colour-space pass 2: no work-order buffer"); return GM_E_ARG; }
      GM_HIP(hipMemsetAsync(d_cls_cnt, 0, 16, stream));
      hipLaunchKernelGGL(k_p2cs_classify, dim3(512), dim3(256), 0, stream, d_work, d_n_work, d_sel, ix.cs_flip, sc.tiebreak_rev ? 1 : 0, d_order, d_cls_cnt);
      if (P.taboo > 0) { if (sc.local) GM_P2CS_GN(true, true); else GM_P2CS_GN(true, false); }
      else { if (sc.local) GM_P2CS_GN(false, true); else GM_P2CS_GN(false, false); }
#undef GM_P2CS_GN
#undef GM_P2CS_G4
      GM_HIP(hipGetLastError());
      return GM_OK;
    }
    if (sc.local) { gm_set_error("colour space: local alignment needs the four-window pass-2 kernel (window of %d does not fit its LDS)", window_len); return GM_E_ARG; }
  }
  if (P.taboo > 0)
  hipLaunchKernelGGL(k_pass2_cs<true>, dim3(grid), dim3(GM_WAVE), lds, stream, ix, sc, P, d_reads, d_initbp, n_reads, read_len, read_words, d_hits, hcap, d_sel,
                     d_work, d_n_work, d_res, d_ops, ops_stride, d_back, back_words, window_len, d_stats, d_xover, d_sel_sidx);
  else
  hipLaunchKernelGGL(k_pass2_cs<false>, dim3(grid), dim3(GM_WAVE), lds, stream, ix, sc, P, d_reads, d_initbp, n_reads, read_len, read_words, d_hits, hcap, d_sel,
                     d_work, d_n_work, d_res, d_ops, ops_stride, d_back, back_words, window_len, d_stats, d_xover, d_sel_sidx);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

// S1's ungapped filter at the seam (ref: common/sw-gapless.c:57-117): call i scores the diagonal of its own genome bitfield (window i starts at word
// woff[i]) through (g_idx, r_idx); with genome_ls the read's first colour is forced against lstocs(letter, init_bp) (:84-94).  One wave per call: the wave
// routine above on a bitfield that starts at the call's first word; the forced first colour enters as the running score's start value.
__global__ void __launch_bounds__(GM_WAVE)
k_sw_gapless_batch(int n, int match, int mismatch, const uint32_t* __restrict__ genome, const uint32_t* __restrict__ genome_ls, const long long* __restrict__ woff,
                   const int* __restrict__ glen, const uint32_t* __restrict__ reads, int read_words, const int* __restrict__ rlen, const int* __restrict__ g_idx,
                   const int* __restrict__ r_idx, const int* __restrict__ initbp, int max_r, int* __restrict__ scores) {
  extern __shared__ __align__(16) uint8_t gl_smem[];
  uint8_t* qr = gl_smem;
  const int lane = threadIdx.x;
  GmScoreDev sc; sc.match = match; sc.mismatch = mismatch;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const uint32_t* g = genome + woff[i]; const uint32_t* rw = reads + (size_t)i * read_words;
    const int rl = rlen[i], gn = glen[i];
    __syncthreads();
    for (int k = lane; k < rl; k += GM_WAVE) qr[k] = (uint8_t)((rw[k >> 3] >> ((k & 7) * 4)) & 0xf);
    __syncthreads();
    long long gi = g_idx[i]; int ri = r_idx[i];
    int head = 0, skip = 0;
    if (genome_ls && gi >= ri) {                       // r_left == 0: the first colour is compared with the letter at g_left and the primer
      const long long gl0 = gi - ri;
      const uint32_t* gls = genome_ls + woff[i];
      const int letter = (int)((gls[gl0 >> 3] >> ((gl0 & 7) * 4)) & 0xf);
      head = (cs_lstocs(letter, initbp[i] & 0xff, (initbp[i] & GM_SEAM_RNA) != 0) == (int)qr[0]) ? match : 0;
      skip = 1;
    }
    // the rest of the diagonal: positions (g_left + skip + k, r_left + skip + k); a start value `head` >= 0 is a first cell of that score
    int best;
    if (!skip) best = sw_gapless_wave(g, 0ull, (long long)gn, qr, rl, gi, ri, sc, lane);
    else {
      // shift the diagonal by one cell and prepend the start value: max-subarray with an initial running score
      const long long g_left = gi - ri + 1; const int r_left = 1;
      const long long room = (long long)gn - g_left;
      const int m = (int)(room < (long long)(rl - r_left) ? room : (long long)(rl - r_left));
      int carry_sum = head, carry_min = 0; best = head;
      for (int k0 = 0; k0 < m; k0 += GM_WAVE) {
        const int k = k0 + lane; int sv = 0;
        if (k < m) { const uint64_t p = (uint64_t)g_left + (uint64_t)k; const uint32_t gc = (g[p >> 3] >> ((p & 7) * 4)) & 0xf; sv = (gc == (uint32_t)qr[r_left + k]) ? match : mismatch; }
        int ps = sv;
        for (int d = 1; d < GM_WAVE; d <<= 1) { const int o = __shfl_up(ps, d); if (lane >= d) ps += o; }
        ps += carry_sum;
        int pm = ps;
        for (int d = 1; d < GM_WAVE; d <<= 1) { const int o = __shfl_up(pm, d); if (lane >= d) pm = min(pm, o); }
        int excl = __shfl_up(pm, 1); if (lane == 0) excl = INT_MAX;
        excl = min(excl, carry_min);
        if (k < m) best = max(best, ps - excl);
        carry_sum = __shfl(ps, GM_WAVE - 1);
        carry_min = min(carry_min, __shfl(pm, GM_WAVE - 1));
      }
      for (int d = 32; d > 0; d >>= 1) best = max(best, __shfl_xor(best, d));
    }
    if (lane == 0) scores[i] = best;
  }
}
int gm_launch_sw_gapless_batch(int n, int match, int mismatch, const uint32_t* d_genome, const uint32_t* d_genome_ls, const long long* d_woff, const int* d_glen,
                               const uint32_t* d_reads, int read_words, const int* d_rlen, const int* d_gidx, const int* d_ridx, const int* d_initbp, int max_r,
                               int* d_scores, hipStream_t stream) {
  if (n == 0) return GM_OK;
  const size_t lds = ((size_t)max_r + 15) & ~(size_t)15;
  hipLaunchKernelGGL(k_sw_gapless_batch, dim3(std::min(n, 256 * 16)), dim3(GM_WAVE), lds, stream, n, match, mismatch, d_genome, d_genome_ls, d_woff, d_glen, d_reads,
                     read_words, d_rlen, d_gidx, d_ridx, d_initbp, max_r, d_scores);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_launch_sw_vector_batch_cs(const GmScoreDev& sc, int n, const uint32_t* d_genome_cs, const uint32_t* d_genome_ls, const long long* d_goff,
                                 const int* d_glen, const uint32_t* d_reads, int read_words, const int* d_rlen, const int* d_initbp, int max_g, int max_r,
                                 int* d_scores, hipStream_t stream) {
  if (n == 0) return GM_OK;
  const size_t lds = ((max_r + 15) & ~15) + 2 * ((max_g + 15) & ~15) + (size_t)max_g * 4 + 64;
  const int grid = std::min(n, 256 * 16);
  hipLaunchKernelGGL(k_sw_vector_batch_cs, dim3(grid), dim3(GM_WAVE), lds, stream, sc, n, d_genome_cs, d_genome_ls, d_goff, d_glen, d_reads, read_words,
                     d_rlen, d_initbp, max_g, max_r, d_scores);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_launch_sw_full_cs_single(const int* cs_params9, const uint32_t* d_genome_ls, long long goff, int glen, const uint32_t* d_read, int rlen, int initbp,
                                int thresh, long long ax, long long ay, int alen, int awidth, int revcmpl, uint32_t* d_back, int* d_out, uint8_t* d_ops,
                                int ops_cap, hipStream_t stream, int local, const int8_t* d_xrow) {
  GmCsDev P; P.match = cs_params9[0]; P.mismatch = cs_params9[1]; P.xover = cs_params9[2]; P.a_go = cs_params9[3]; P.a_ge = cs_params9[4];
  P.b_go = cs_params9[5]; P.b_ge = cs_params9[6]; P.anchor_width = cs_params9[7]; P.taboo = cs_params9[8];
  const size_t lds = 5 * (size_t)((rlen + 15) & ~15) + ((glen + 15) & ~15) + (size_t)glen * 48 + 64;
  static GmLdsLimit lim_configured; size_t& configured = lim_configured.cur();
  if (lds > 160 * 1024) { gm_set_error("sw_full_cs: window of %d does not fit LDS", glen); return GM_E_ARG; }
  if (lds > 48 * 1024 && lds > configured) { GM_HIP(hipFuncSetAttribute((const void*)k_sw_full_cs_single, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); configured = lds; }
  hipLaunchKernelGGL(k_sw_full_cs_single, dim3(1), dim3(GM_WAVE), lds, stream, P, d_genome_ls, goff, glen, d_read, rlen, initbp, thresh, ax, ay, alen, awidth,
                     revcmpl, d_back, d_out, d_ops, ops_cap, local, d_xrow);
  GM_HIP(hipGetLastError());
  return GM_OK;
}
