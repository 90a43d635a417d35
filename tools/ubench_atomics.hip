// Microbenchmark: random-access update rates on MI355X for table sizes in/out of the caches.
// Measures what bounds the build kernel: plain 16-B probe loads, agent-scope (memory-side)
// atomics, and workgroup-scope (XCD-L2) atomics.  hipcc --offload-arch=gfx950 -O3 ubench_atomics.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// MODE 0: load16 only; 1: agent atomic add; 2: workgroup atomic add; 3: load16 + agent atomic;
// 4: load16 + workgroup atomic; 5: agent atomic with return
template <int MODE, int PER>
__global__ __launch_bounds__(256) void k(uint64_t *tab, uint64_t nrec, uint64_t n, uint64_t seed, uint64_t *sink, int xcd_local)
{
  uint64_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * PER) {
    uint64_t idx[PER];
    ulonglong2 v[PER];
#pragma unroll
    for (int p = 0; p < PER; p++) {
      uint64_t r = mix((i0 + p * stride) ^ seed);
      uint64_t slot = r % nrec;
      if (xcd_local) {  // keep each XCD (blockIdx % 8) in its own eighth of the table
        const uint64_t per = nrec / 8;
        slot = (blockIdx.x % 8) * per + r % per;
      }
      idx[p] = slot;
    }
    if (MODE == 0 || MODE == 3 || MODE == 4) {
#pragma unroll
      for (int p = 0; p < PER; p++) v[p] = *reinterpret_cast<const ulonglong2 *>(tab + 2 * idx[p]);
#pragma unroll
      for (int p = 0; p < PER; p++) acc += v[p].x;
    }
#pragma unroll
    for (int p = 0; p < PER; p++) {
      uint64_t *a = tab + 2 * idx[p] + 1;
      if (MODE == 1 || MODE == 3) __hip_atomic_fetch_add(a, 256ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 2 || MODE == 4) __hip_atomic_fetch_add(a, 256ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 5) acc += __hip_atomic_fetch_add(a, 256ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

template <int MODE>
static void run(const char *name, uint64_t *tab, uint64_t nrec, uint64_t n, uint64_t *sink, int xcd_local)
{
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipMemset(tab, 0, nrec * 16);
  k<MODE, 4><<<2048, 256>>>(tab, nrec, n / 8, 1, sink, xcd_local);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE, 4><<<2048, 256>>>(tab, nrec, n, 7, sink, xcd_local);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // verify sum for atomic modes
  printf("  %-28s %8.2f ms  %7.2f G/s%s\n", name, ms, n / ms / 1e6, xcd_local ? "  [xcd-local]" : "");
}

int main()
{
  uint64_t *sink; hipMalloc(&sink, 8);
  const uint64_t n = 1ull << 29;
  for (uint64_t mb : {32ull, 128ull, 2048ull, 16384ull}) {
    const uint64_t nrec = mb * 1024 * 1024 / 16;
    uint64_t *tab; if (hipMalloc(&tab, nrec * 16) != hipSuccess) { printf("alloc fail\n"); return 1; }
    printf("table %llu MB, %llu random updates\n", (unsigned long long)mb, (unsigned long long)n);
    run<0>("load16", tab, nrec, n, sink, 0);
    run<1>("atomic agent", tab, nrec, n, sink, 0);
    run<5>("atomic agent (returning)", tab, nrec, n, sink, 0);
    run<2>("atomic workgroup", tab, nrec, n, sink, 0);
    run<2>("atomic workgroup", tab, nrec, n, sink, 1);
    run<3>("load16 + atomic agent", tab, nrec, n, sink, 0);
    run<4>("load16 + atomic workgroup", tab, nrec, n, sink, 0);
    run<4>("load16 + atomic workgroup", tab, nrec, n, sink, 1);
    hipFree(tab);
  }
  return 0;
}
