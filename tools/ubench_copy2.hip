// Microbenchmark: why does tools/ubench_copy.hip top out at 5.15 TB/s when MI355X_MICROARCH.md quotes 6.29 TB/s
// for a float4 copy?  Variants of a 16-byte-per-lane copy: access pattern (grid-stride over the whole buffer vs a
// contiguous chunk per block), non-temporal loads / stores, buffer size (in / out of the 256 MiB Infinity Cache,
// TLB reach), loads in flight.  hipcc --offload-arch=gfx950 -O3 tools/ubench_copy2.hip -o build/ubench_copy2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));

// PAT 0: grid-stride (element i0 + u * gridDim * 256); PAT 1: block b copies the contiguous elements
// [b * per_block, (b + 1) * per_block), a wave's UN loads are UN consecutive 1 KiB rows
template <int PAT, int UN, int NT>
__global__ __launch_bounds__(256) void kcopy(const f4 *__restrict__ src, f4 *__restrict__ dst, uint64_t n)
{
  if (PAT == 0) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i0 = (uint64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += stride * UN) {
      f4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; u++) { const uint64_t i = i0 + u * stride; if (i < n) v[u] = (NT & 1) ? __builtin_nontemporal_load(src + i) : src[i]; }
#pragma unroll
      for (int u = 0; u < UN; u++) { const uint64_t i = i0 + u * stride; if (i < n) { if (NT & 2) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; } }
    }
  } else {
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * UN) {
      f4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; u++) { const uint64_t i = i0 + u * 256; if (i < hi) v[u] = (NT & 1) ? __builtin_nontemporal_load(src + i) : src[i]; }
#pragma unroll
      for (int u = 0; u < UN; u++) { const uint64_t i = i0 + u * 256; if (i < hi) { if (NT & 2) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; } }
    }
  }
}
template <int PAT, int UN, int NT> static double run(f4 *a, f4 *b, uint64_t n, int grid)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kcopy<PAT, UN, NT><<<grid, 256>>>(a, b, n); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) kcopy<PAT, UN, NT><<<grid, 256>>>(a, b, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double tbs = 2.0 * n * 16 / ms / 1e9;
  printf("  pattern %s unroll %d nt %d grid %6d: %8.3f ms  %.2f TB/s\n", PAT ? "block-contiguous" : "grid-stride     ", UN, NT, grid, ms, tbs);
  return tbs;
}
int main()
{
  for (uint64_t gib4 : {1ull, 4ull, 32ull, 128ull}) {  // buffer size in quarters of a GiB: 256 MiB, 1, 8, 32 GiB
    const uint64_t n = (gib4 << 28) / 16;
    f4 *a, *b;
    if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess) { printf("alloc failed at %llu MiB\n", (u64)(gib4 * 256)); break; }
    hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16);
    printf("buffers of %llu MiB (read one, write the other)\n", (u64)(gib4 * 256));
    for (int grid : {2048, 8192, 65536}) {
      run<0, 4, 0>(a, b, n, grid);
      run<0, 8, 0>(a, b, n, grid);
      run<1, 4, 0>(a, b, n, grid);
      run<1, 8, 0>(a, b, n, grid);
    }
    run<0, 8, 1>(a, b, n, 8192); run<0, 8, 2>(a, b, n, 8192); run<0, 8, 3>(a, b, n, 8192);
    run<1, 8, 1>(a, b, n, 8192); run<1, 8, 2>(a, b, n, 8192); run<1, 8, 3>(a, b, n, 8192);
    run<1, 16, 0>(a, b, n, 8192); run<1, 16, 3>(a, b, n, 8192);
    hipFree(a); hipFree(b);
  }
  return 0;
}
