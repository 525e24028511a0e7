// gm_post.hip -- colour-space post_sw on the device (ref: common/sw-post.c:111-758) for reads without quality values.
//
//   k_post_sw_cs     one thread per pass-2 result: load_local_vectors, do_forwards, do_backwards, post_traceback + fix_base_calls, get_posterior
//
// The 16-state forward-backward of cs_post_sw (gm_host.hip) in the same operation order, in doubles, with contraction off -- what differs from
// the host routine is the exp / log implementation (ocml here, glibc there; both within 1 ulp of the true value).  The sums are well conditioned
// (every column is rescaled by its minimum), so `total` carries an absolute error of ~1e-12 after 50 columns; an output integer (AS, a base
// quality, MAPQ, the Z tags) could differ from the host's only when the value rounded lies that close to a rounding boundary -- those are caught on the
// host and redone there (below).  GM_POST_SW_HOST=1 keeps the host routine throughout, and FASTQ input (per-colour error
// rates from the QVs, base qualities) always takes it.  A letter call between two (nearly) equal posteriors is decided by the last bits too:
// the kernel flags those results (valid = 2) and the host routine redoes them -- as it does a result with a base quality (reads with QVs) whose
// truncation to an integer falls within 1e-7 of its boundary.
// Reads with quality values (round 3): the per-colour error rates come from a table the host computed with its own libm (251 entries: the same doubles
// the host routine uses), so again only exp / log of the recursion differ.
//
// What the kernel leaves: GmPostRes per result (posterior, the match / mismatch / crossover counts), and the re-called letters written into the spare
// bits of the result's backtrace bytes -- letter in bits 4-5 of bt[t], lower-case flag in bit 6; the type nibble, sw_full_cs's crossover mark (bit 7) and
// the letter codes stay as they were -- so that the host's cs_alignment_strings yields the final qralign directly AND the host routine can still redo
// any result from sw_full_cs's own record.  The host does that whenever a value it is about to round (AS, MAPQ, the Z tags) lies within 1e-7 of its
// rounding boundary (Finalizer::post_sw / finalize_read, gm_host.hip): there the last bits of exp / log would decide an output byte.
#include "gm_common.h"
#include "gm_internal.h"

// column descriptor: bits 0-2 let + 2 (0: insertion, 1: no state matches, 2-5: A C G T), 3-4 colour, 5 which error rate, 8-10 crt (filled by the backward sweep),
// 12-14 the letter sw_full_cs had called (7: none), 16-23 + bit 24: the colour's quality value (reads with QVs)
__device__ __forceinline__ double k_prior(const GmCsPostDev& K, uint32_t info, int st) {                    // nodePrior, ref: sw-post.c:111-138
  const int let = (int)(info & 7u) - 2, col = (int)((info >> 3) & 3u), which = (int)((info >> 5) & 1u);
  const int l = (st >> 2) & 3, r = st & 3;
  double val = 0;
  if (let != -2) val = val - ((r == let) ? K.let_m : K.let_x);
  double cm = K.col_m[which], cx = K.col_x[which];
  if (info & (1u << 24)) { const uint32_t q = (info >> 16) & 0xFFu; cm = K.qtab[2 * q]; cx = K.qtab[2 * q + 1]; }     // the colour's own error rate (reads with QVs)
  val = val - (((l ^ r) == col) ? cm : cx);
  return val;
}

__global__ void __launch_bounds__(64)
k_post_sw_cs(GmCsPostDev K, const uint32_t* __restrict__ reads, const uint8_t* __restrict__ initbp, int read_len, int read_words,
             const GmFullRes* __restrict__ res, uint8_t* __restrict__ ops, int ops_stride, const uint32_t* __restrict__ n_work_p, uint32_t res_cap,
             GmPostRes* __restrict__ post, double* __restrict__ fwbuf, uint32_t* __restrict__ infobuf) {
#pragma clang fp contract(off)
  const uint32_t T = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_work = min(*n_work_p, res_cap);
  const int ops_half = ops_stride / 2;
  // scratch, interleaved over the threads: fwbuf[(c * 17 + k) * T + tid] (k = 16: the column's scale fs), infobuf[c * T + tid]
  for (uint32_t w = tid; w < n_work; w += T) {
    const GmFullRes R = res[w];
    GmPostRes out; out.posterior = 0; out.cs_match = out.cs_mismatch = out.cs_xover = 0; out.valid = 0;
    if (R.score > 0) {
      uint8_t* bt = ops + (size_t)R.ops_off; uint8_t* codes = bt + ops_half;
      const int n = min(R.n_ops, ops_half);
      const uint32_t* rw = reads + (size_t)R.read_idx * read_words;
      const int init_bp = (int)initbp[R.read_idx];
      auto colour = [&](int j) { return (int)((rw[j >> 3] >> ((j & 7) * 4)) & 0xf); };
      // ---- load_local_vectors, ref: sw-post.c:448-528 ----
      int start_run = 0, len = 0, min_qv = 255;
      const uint8_t* qvr = K.qv ? K.qv + (size_t)R.read_idx * read_len : nullptr;
      { int j; for (j = 0; j < R.read_start; j++) { const int c = colour(j); if (c == 15) { start_run = 15; min_qv = 0; break; } start_run ^= c; if (qvr) min_qv = min(min_qv, (int)qvr[j]); } }
      { int j = R.read_start;
        for (int t = 0; t < n; t++) {
          const int type = bt[t] & 0x0f; if (type == 1) continue;                     // deletion: no read position
          const bool ins = type >= 2 && type <= 5;
          const int d = codes[t] >> 4;
          const int let = ins ? -2 : (d < 4 ? d : -1);
          const int cc = j < read_len ? colour(j) : 15; int col, which;
          if ((len == 0 && start_run == 15) || cc == 15) { col = 0; which = 1; } else { col = cc ^ (len == 0 ? start_run : 0); which = 0; }
          uint32_t word = (uint32_t)(let + 2) | ((uint32_t)col << 3) | ((uint32_t)which << 5);
          int bc = codes[t] & 15;
          if (bc == 15 && !ins) bc = d;                             // an unknown read letter is shown as the genome's (ref: sw-full-cs.c pretty_print), and THAT is what post_sw reads as the base call
          word |= (uint32_t)(bc < 4 ? bc : 7) << 12;
          if (qvr && which == 0 && j < read_len) {                   // ref: sw-post.c:486-491 (the first column takes the smallest QV of the skipped colours and its own)
            const int q = len == 0 ? min(min_qv, (int)qvr[j]) : (int)qvr[j];
            word |= ((uint32_t)q << 16) | (1u << 24);
          }
          infobuf[(size_t)len * T + tid] = word;
          len++; j++;
        } }
      if (len > 0) {
        double fw[16], fs;
        // ---- do_forwards, ref: sw-post.c:317-360 ----
        { const uint32_t info = infobuf[tid];
          fs = 999999999;
          for (int j = 0; j < 16; j++) { if (((j >> 2) & 3) == init_bp) { fw[j] = k_prior(K, info, j); fs = (fs < fw[j]) ? fs : fw[j]; } else fw[j] = HUGE_VAL; }
          for (int j = 0; j < 16; j++) { fw[j] -= fs; fwbuf[(size_t)j * T + tid] = fw[j]; }
          fwbuf[(size_t)16 * T + tid] = fs; }
        for (int i = 1; i < len; i++) {
          const uint32_t info = infobuf[(size_t)i * T + tid];
          double e[16], lg[4];
          for (int k = 0; k < 16; k++) e[k] = exp(-1 * fw[k]);
          for (int l = 0; l < 4; l++) { double sum = 0; for (int k = l; k < 16; k += 4) sum += e[k]; lg[l] = log(sum); }
          double cfs = 999999999;
          for (int j = 0; j < 16; j++) { fw[j] = k_prior(K, info, j) - lg[(j >> 2) & 3]; cfs = (cfs < fw[j]) ? cfs : fw[j]; }
          for (int j = 0; j < 16; j++) { fw[j] -= cfs; fwbuf[((size_t)i * 17 + j) * T + tid] = fw[j]; }
          cfs += fs; fs = cfs;
          fwbuf[((size_t)i * 17 + 16) * T + tid] = fs;
        }
        double total;
        { double val = 0; for (int j = 0; j < 16; j++) val += exp(-1 * fw[j]); total = -log(val) + fs; }
        // ---- do_backwards (ref: sw-post.c:269-315) with the posterior of every column taken on the way (post_traceback, ref: sw-post.c:183-212) ----
        double bw[16], bs; bool tie = false;
        { bs = 999999999; for (int j = 0; j < 16; j++) { bw[j] = 0; bs = (bs < bw[j]) ? bs : bw[j]; } for (int j = 0; j < 16; j++) bw[j] -= bs; }
        for (int i = len - 1; i >= 0; i--) {
          const uint32_t info = infobuf[(size_t)i * T + tid];
          const double cfs = fwbuf[((size_t)i * 17 + 16) * T + tid];
          double p4[4] = {0, 0, 0, 0};
          for (int st = 0; st < 16; st++) {
            const double a = fwbuf[((size_t)i * 17 + st) * T + tid] + bw[st] + cfs + bs - total;
            p4[st & 3] += exp(-1 * a);
          }
          int crt = 0; for (int b = 1; b < 4; b++) if (p4[b] > p4[crt]) crt = b;
          // A call between two letters whose posteriors are (nearly) equal -- an inserted base next to a colour error has two explanations of the very
          // same probability -- is decided by the last bits of exp / log: such a result goes to the host routine (0.1 % of the results).
          for (int b = 0; b < 4; b++) if (b != crt && p4[b] >= p4[crt] * (1.0 - 1e-9)) tie = true;
          infobuf[(size_t)i * T + tid] = info | ((uint32_t)crt << 8);
          if (K.bq) {                                                 // get_base_qualities, ref: sw-post.c:568-586 (qv_from_pr_corr, util.h:267-283; at most 40)
            const int bc = (int)((info >> 12) & 7u); int tq = 0;
            if (bc < 4) {
              const double pr_err = 1 - p4[bc];
              if (fabs(pr_err - .99999999) < 1e-12 || fabs(pr_err - 1E-25) < 1e-34) tie = true;        // at one of the two cut-offs: the host decides
              if (pr_err > .99999999) tq = 0; else if (pr_err < 1E-25) tq = 250;
              else { const double v = -10.0 * log(pr_err) / log(10.0); if (v < 41.5 && fabs(v - rint(v)) < 1e-7) tie = true; tq = (int)v; }   // a truncation at its boundary: the host decides
            }
            if (tq > 40) tq = 40;
            K.bq[(size_t)w * read_len + i] = (uint8_t)(33 + tq);
          }
          if (i > 0) {
            double e[16], nl[4];
            for (int k = 0; k < 16; k++) { const double a = k_prior(K, info, k) + bw[k]; e[k] = exp(-1 * a); }
            for (int r = 0; r < 4; r++) { double sum = 0; for (int k = 4 * r; k < 4 * r + 4; k++) sum += e[k]; nl[r] = -log(sum); }
            double cbs = 999999999;
            for (int j = 0; j < 16; j++) { bw[j] = nl[j & 3]; cbs = (cbs < bw[j]) ? cbs : bw[j]; }
            for (int j = 0; j < 16; j++) bw[j] -= cbs;
            cbs += bs; bs = cbs;
          }
        }
        // ---- fix_base_calls, ref: sw-post.c:531-565: the re-called letters go into the op record ----
        if (!tie) { int j = 0, prev_base = init_bp;
          for (int t = 0; t < n; t++) {
            const int type = bt[t] & 0x0f; if (type == 1) continue;
            const uint32_t info = infobuf[(size_t)j * T + tid];
            const int crt = (int)((info >> 8) & 3u), col = (int)((info >> 3) & 3u);
            const bool lower = (prev_base ^ crt) != col;
            if (lower) out.cs_xover++;
            const int d = codes[t] >> 4;
            if (!(type >= 2 && type <= 5)) { if (d == crt) out.cs_match++; else out.cs_mismatch++; }
            bt[t] = (uint8_t)((bt[t] & 0x8f) | (crt << 4) | (lower ? 0x40 : 0));          // the re-call rides in the spare bits: sw_full_cs's own letters and marks stay
            prev_base = crt; j++;
          } }
        // ---- get_posterior, ref: sw-post.c:589-612 ----
        if (!tie) { double r = exp(-total); bool prev_ins = false, prev_del = false;
          for (int t = 0; t < n; t++) {
            const int type = bt[t] & 0x0f; const bool ins = type >= 2 && type <= 5, del = type == 1;
            if (ins) { r *= K.pr_ins_extend; if (!prev_ins) r *= K.pr_ins_open; }
            else if (del) { r *= K.pr_del_extend; if (!prev_del) r *= K.pr_del_open; }
            prev_ins = ins; prev_del = del;
          }
          out.posterior = r; }
        if (tie) out.valid = 2;                                   // the host's cs_post_sw takes it from here
      }
      if (out.valid == 0) out.valid = 1;
    }
    post[w] = out;
  }
}

int gm_launch_post_sw_cs(const GmCsPostDev& K, const uint32_t* d_reads, const uint8_t* d_initbp, int read_len, int read_words, const GmFullRes* d_res, uint8_t* d_ops,
                         int ops_stride, const uint32_t* d_n_work, uint32_t res_cap, GmPostRes* d_post, double* d_fw, uint32_t* d_info, int threads, hipStream_t stream) {
  hipLaunchKernelGGL(k_post_sw_cs, dim3(threads / 64), dim3(64), 0, stream, K, d_reads, d_initbp, read_len, read_words, d_res, d_ops, ops_stride, d_n_work, res_cap,
                     d_post, d_fw, d_info);
  if (hipGetLastError() != hipSuccess) return GM_E_NODEVICE;
  return GM_OK;
}
