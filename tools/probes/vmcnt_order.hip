// Probe: do hand-issued global_load_dwordx4 (saddr form) return in issue order, i.e. is "s_waitcnt vmcnt(Q-1)" enough for the oldest of Q loads?
// a[i] = i; every wave keeps Q loads in flight to pseudo-random places (mix of cache hits and HBM misses) and checks each load after vmcnt(Q-1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
#define Q 6
__global__ void __launch_bounds__(1024) k(const uint32_t* __restrict__ a, uint64_t n, int iters, unsigned long long* bad, int mode) {
  const int lane = threadIdx.x & 63;
  uint32_t seed = (blockIdx.x * 1024 + threadIdx.x) >> 6; seed = seed * 2654435761u + 12345u;
  auto nxt = [&]() -> uint64_t {   // wave-uniform pseudo-random chunk start
    seed = seed * 1664525u + 1013904223u;
    uint32_t r = __builtin_amdgcn_readfirstlane(seed);
    uint64_t off = (mode & 1) ? (uint64_t)(r % 4096u) * 256u : ((uint64_t)r * 977u) % (n - 1024);   // mode 1: small hot set (cache hits)
    if ((mode & 2) && (r & 8)) off = (uint64_t)(r % 4096u) * 256u;                                  // mode 2: mix
    return off;
  };
  u32x4 v[Q]; uint64_t o[Q];
  unsigned long long nbad = 0;
#pragma unroll
  for (int q = 0; q < Q; q++) { o[q] = nxt(); const uint32_t e = lane * 12; const uint32_t* src = a + o[q]; asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v[q]) : "v"(e), "s"(src) : "memory"); }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int q = 0; q < Q; q++) {
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[q]) : "n"(Q - 1) : "memory");
      const uint32_t want = (uint32_t)(o[q] + lane * 3);
      if (v[q].x != want || v[q].y != want + 1 || v[q].z != want + 2 || v[q].w != want + 3) nbad++;
      o[q] = nxt(); const uint32_t e = lane * 12; const uint32_t* src = a + o[q];
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v[q]) : "v"(e), "s"(src) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (nbad) atomicAdd(bad, nbad);
}
int main() {
  const uint64_t n = 1ull << 28;   // 1 GiB of uint32
  uint32_t* a; unsigned long long* bad;
  hipMalloc(&a, n * 4); hipMalloc(&bad, 8);
  std::vector<uint32_t> h(1 << 20);
  for (uint64_t base = 0; base < n; base += h.size()) { for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(base + i); hipMemcpy(a + base, h.data(), h.size() * 4, hipMemcpyHostToDevice); }
  for (int mode = 0; mode < 4; mode++) {
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, a, n, 2000, bad, mode);
    unsigned long long b = 0; hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost);
    printf("mode %d: mismatching lane-loads %llu of %llu (%s)\n", mode, b, 256ull * 1024 * 2000 * Q, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
