// Microbenchmark: what streaming rate does this part give a hand-written kernel?  copy (R + W),
// read-only and write-only over buffers far larger than the 256 MiB Infinity Cache, 16-byte accesses,
// 8 loads in flight per lane, persistent grid.  hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
template <int MODE, int UN>
__global__ __launch_bounds__(256) void k(const ulonglong2 *src, ulonglong2 *dst, uint64_t n, u64 *sink)
{
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  u64 acc = 0;
  for (uint64_t i0 = (uint64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += stride * UN) {
    ulonglong2 v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) { const uint64_t i = i0 + u * stride; v[u] = (MODE != 2 && i < n) ? src[i] : make_ulonglong2(i, i); }
#pragma unroll
    for (int u = 0; u < UN; u++) { const uint64_t i = i0 + u * stride; if (MODE == 1) acc += v[u].x ^ v[u].y; else if (i < n) dst[i] = v[u]; }
  }
  if (MODE == 1 && acc == 0x1234567) sink[0] = acc;
}
template <int MODE, int UN> static void run(const char *name, ulonglong2 *a, ulonglong2 *b, uint64_t n, u64 *sink, int grid)
{
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, UN><<<grid, 256>>>(a, b, n, sink); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) k<MODE, UN><<<grid, 256>>>(a, b, n, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (MODE == 0 ? 2.0 : 1.0) * n * 16;
  printf("%-28s grid %5d unroll %d: %7.3f ms  %.2f TB/s\n", name, grid, UN, ms, bytes / ms / 1e9);
}
int main()
{
  const uint64_t n = (8ull << 30) / 16;  // 8 GiB per buffer
  ulonglong2 *a, *b; u64 *sink;
  hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&sink, 8);
  hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16);
  for (int grid : {1024, 2048, 4096, 8192}) {
    run<0, 4>("copy (read + write)", a, b, n, sink, grid);
    run<0, 8>("copy (read + write)", a, b, n, sink, grid);
  }
  run<1, 8>("read only", a, b, n, sink, 2048);
  run<1, 8>("read only", a, b, n, sink, 8192);
  run<2, 8>("write only", a, b, n, sink, 2048);
  run<2, 8>("write only", a, b, n, sink, 8192);
  return 0;
}
