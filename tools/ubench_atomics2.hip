// Microbenchmark 2: what limits random device atomics on MI355X?  (see ubench_atomics.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
// MODE 0: u64 atomic add; 1: u32 atomic add; 2: plain 16B store; 3: plain 8B store; 4: load16 then plain store16 (RMW, racy)
// 5: u64 atomic, 4 lanes share a 64B line (4 consecutive records); 6: u64 atomic or; 7: u64 atomic, whole wave in 1 KiB window
// 8: u32 atomic, 16 lanes share a 64B line
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *tab, uint64_t nrec, uint64_t n, uint64_t seed, uint64_t *sink)
{
  uint64_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t r = mix(i ^ seed);
    uint64_t slot = r % nrec;
    if (MODE == 5) { uint64_t g = mix((i >> 2) ^ seed) % (nrec / 4); slot = g * 4 + (i & 3); }
    if (MODE == 7) { uint64_t g = mix((i >> 6) ^ seed) % (nrec / 64); slot = g * 64 + (i & 63); }
    uint64_t *rec = tab + 2 * slot;
    if (MODE == 0 || MODE == 5 || MODE == 7) __hip_atomic_fetch_add(rec + 1, 256ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) __hip_atomic_fetch_add((uint32_t *)(rec + 1), 256u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 8) { uint64_t g = mix((i >> 4) ^ seed) % (nrec / 4); uint32_t *p = (uint32_t *)(tab + 8 * g) + (i & 15);
                     __hip_atomic_fetch_add(p, 256u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (MODE == 6) __hip_atomic_fetch_or(rec + 1, 1ULL << (r >> 58), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 2) *reinterpret_cast<ulonglong2 *>(rec) = make_ulonglong2(r, i);
    if (MODE == 3) rec[1] = r;
    if (MODE == 4) { ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(rec); v.y += 256; *reinterpret_cast<ulonglong2 *>(rec) = v; }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

template <int MODE>
static void run(const char *name, uint64_t *tab, uint64_t nrec, uint64_t n, uint64_t *sink, int grid)
{
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<grid, 256>>>(tab, nrec, n / 8, 1, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE><<<grid, 256>>>(tab, nrec, n, 7, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("  %-44s grid %5d %8.2f ms  %7.2f G/s\n", name, grid, ms, n / ms / 1e6);
}

int main()
{
  uint64_t *sink; hipMalloc(&sink, 8);
  const uint64_t n = 1ull << 28;
  for (uint64_t mb : {128ull, 16384ull}) {
    const uint64_t nrec = mb * 1024 * 1024 / 16;
    uint64_t *tab; if (hipMalloc(&tab, nrec * 16) != hipSuccess) { printf("alloc fail\n"); return 1; }
    hipMemset(tab, 0, nrec * 16);
    printf("table %llu MB, %llu random updates\n", (unsigned long long)mb, (unsigned long long)n);
    for (int grid : {256, 1024, 2048, 8192}) run<0>("u64 atomic add", tab, nrec, n, sink, grid);
    run<1>("u32 atomic add", tab, nrec, n, sink, 2048);
    run<6>("u64 atomic or", tab, nrec, n, sink, 2048);
    run<5>("u64 atomic add, 4 lanes per 64B line", tab, nrec, n, sink, 2048);
    run<7>("u64 atomic add, wave in 1KiB window", tab, nrec, n, sink, 2048);
    run<8>("u32 atomic add, 16 lanes per 64B line", tab, nrec, n, sink, 2048);
    run<2>("plain store 16B", tab, nrec, n, sink, 2048);
    run<3>("plain store 8B", tab, nrec, n, sink, 2048);
    run<4>("load16 + store16 (non-atomic RMW)", tab, nrec, n, sink, 2048);
    hipFree(tab);
  }
  return 0;
}
