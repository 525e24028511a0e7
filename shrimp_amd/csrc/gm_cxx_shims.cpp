// gm_cxx_shims.cpp -- the reference's own (C++-linkage) names for the kernel seams.
//
// SHRiMP 2.2.3 compiles its .c files as C++ and every `extern "C"` in its headers is commented out (ref: common/util.h:8-10), so gmapper's objects
// reference Itanium-mangled symbols: _Z9sw_vectorPjiiS_iS_ib and friends.  libgmapper_hip.so exports the seams with C linkage (include/gmapper_hip.h);
// this file adds the mangled twins, each a one-line forward, so that the reference's unmodified objects link against the library once
// common/sw-vector.o, sw-full-ls.o, sw-full-cs.o and sw-post.o are dropped from the link line (tests/test_abi.py links them).
// Declarations follow common/sw-vector.h:3-6, sw-gapless.h:11-14, sw-full-ls.h:9-13, sw-full-cs.h:7-11, sw-post.h:8-12; the two structs are only named here (their
// layouts are gm_sw_full_results / gm_anchor in the public header, field for field the reference's).
#include <stdint.h>
struct sw_full_results;
struct anchor;

extern "C" {
int  gmc_sw_vector_setup(int, int, int, int, int, int, int, int, int, bool) __asm__("sw_vector_setup");
int  gmc_sw_vector_cleanup(void) __asm__("sw_vector_cleanup");
void gmc_sw_vector_stats(uint64_t*, uint64_t*, double*) __asm__("sw_vector_stats");
int  gmc_sw_vector(uint32_t*, int, int, uint32_t*, int, uint32_t*, int, bool) __asm__("sw_vector");
int  gmc_sw_gapless_setup(int, int, bool) __asm__("sw_gapless_setup");
void gmc_sw_gapless_stats(uint64_t*, uint64_t*, uint64_t*) __asm__("sw_gapless_stats");
int  gmc_sw_gapless(uint32_t*, int, uint32_t*, int, int, int, uint32_t*, int, bool) __asm__("sw_gapless");
int  gmc_sw_full_ls_setup(int, int, int, int, int, int, int, int, bool, int) __asm__("sw_full_ls_setup");
int  gmc_sw_full_ls_cleanup(void) __asm__("sw_full_ls_cleanup");
void gmc_sw_full_ls_stats(uint64_t*, uint64_t*, double*) __asm__("sw_full_ls_stats");
void gmc_sw_full_ls(uint32_t*, int, int, uint32_t*, int, int, int, struct sw_full_results*, bool, struct anchor*, int, int) __asm__("sw_full_ls");
int  gmc_sw_full_cs_setup(int, int, int, int, int, int, int, int, int, bool, int, int) __asm__("sw_full_cs_setup");
int  gmc_sw_full_cs_cleanup(void) __asm__("sw_full_cs_cleanup");
void gmc_sw_full_cs_stats(uint64_t*, uint64_t*, double*) __asm__("sw_full_cs_stats");
void gmc_sw_full_cs(uint32_t*, int, int, uint32_t*, int, int, int, struct sw_full_results*, bool, bool, struct anchor*, int, int, int*) __asm__("sw_full_cs");
int  gmc_post_sw_setup(int, double, double, double, double, double, double, bool, bool, int, int, bool) __asm__("post_sw_setup");
int  gmc_post_sw_cleanup(void) __asm__("post_sw_cleanup");
int  gmc_post_sw_stats(uint64_t*, uint64_t*, double*) __asm__("post_sw_stats");
void gmc_post_sw(uint32_t*, int, char*, struct sw_full_results*) __asm__("post_sw");
}

#define GM_EXPORT __attribute__((visibility("default")))
GM_EXPORT int sw_vector_setup(int a, int b, int c, int d, int e, int f, int g, int h, int i, bool j) { return gmc_sw_vector_setup(a, b, c, d, e, f, g, h, i, j); }
GM_EXPORT int sw_vector_cleanup(void) { return gmc_sw_vector_cleanup(); }
GM_EXPORT void sw_vector_stats(uint64_t* a, uint64_t* b, double* c) { gmc_sw_vector_stats(a, b, c); }
GM_EXPORT int sw_vector(uint32_t* a, int b, int c, uint32_t* d, int e, uint32_t* f, int g, bool h) { return gmc_sw_vector(a, b, c, d, e, f, g, h); }
GM_EXPORT int sw_gapless_setup(int a, int b, bool c) { return gmc_sw_gapless_setup(a, b, c); }
GM_EXPORT void sw_gapless_stats(uint64_t* a, uint64_t* b, uint64_t* c) { gmc_sw_gapless_stats(a, b, c); }
GM_EXPORT int sw_gapless(uint32_t* a, int b, uint32_t* c, int d, int e, int f, uint32_t* g, int h, bool i) { return gmc_sw_gapless(a, b, c, d, e, f, g, h, i); }
GM_EXPORT int sw_full_ls_setup(int a, int b, int c, int d, int e, int f, int g, int h, bool i, int j) { return gmc_sw_full_ls_setup(a, b, c, d, e, f, g, h, i, j); }
GM_EXPORT int sw_full_ls_cleanup(void) { return gmc_sw_full_ls_cleanup(); }
GM_EXPORT void sw_full_ls_stats(uint64_t* a, uint64_t* b, double* c) { gmc_sw_full_ls_stats(a, b, c); }
GM_EXPORT void sw_full_ls(uint32_t* a, int b, int c, uint32_t* d, int e, int f, int g, struct sw_full_results* h, bool i, struct anchor* j, int k, int l) {
  gmc_sw_full_ls(a, b, c, d, e, f, g, h, i, j, k, l);
}
GM_EXPORT int sw_full_cs_setup(int a, int b, int c, int d, int e, int f, int g, int h, int i, bool j, int k, int l) { return gmc_sw_full_cs_setup(a, b, c, d, e, f, g, h, i, j, k, l); }
GM_EXPORT int sw_full_cs_cleanup(void) { return gmc_sw_full_cs_cleanup(); }
GM_EXPORT void sw_full_cs_stats(uint64_t* a, uint64_t* b, double* c) { gmc_sw_full_cs_stats(a, b, c); }
GM_EXPORT void sw_full_cs(uint32_t* a, int b, int c, uint32_t* d, int e, int f, int g, struct sw_full_results* h, bool i, bool j, struct anchor* k, int l, int m, int* n) {
  gmc_sw_full_cs(a, b, c, d, e, f, g, h, i, j, k, l, m, n);
}
GM_EXPORT int post_sw_setup(int a, double b, double c, double d, double e, double f, double g, bool h, bool i, int j, int k, bool l) { return gmc_post_sw_setup(a, b, c, d, e, f, g, h, i, j, k, l); }
GM_EXPORT int post_sw_cleanup() { return gmc_post_sw_cleanup(); }
GM_EXPORT int post_sw_stats(uint64_t* a, uint64_t* b, double* c) { return gmc_post_sw_stats(a, b, c); }
GM_EXPORT void post_sw(uint32_t* a, int b, char* c, struct sw_full_results* d) { gmc_post_sw(a, b, c, d); }
