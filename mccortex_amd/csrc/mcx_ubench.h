// mcx_ubench.h -- the two ceilings of the device that SURVEY.md 8(d) wants measured on the box, in the same
// run as the build they are quoted against: (i) streaming copy GB/s, (ii) random 64-byte-sector RMW rate over a
// working set equal to the table.  Included by mcx_api.hip; entry points mcx_ubench_stream / mcx_ubench_random_rmw
// (include/mcx_gpu.h).  Measurement support only: no build path calls them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcx {

// MODE 0: copy (16-byte loads and stores), 1: read only, 2: write only.  Block b takes a contiguous chunk;
// a lane keeps UN 16-byte accesses in flight.
template <int MODE, int UN>
__global__ __launch_bounds__(256) void k_ubench_stream(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ dst, uint64_t n,
                                                       unsigned long long *sink)
{
  const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  unsigned long long acc = 0;
  for (uint64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * UN) {
    ulonglong2 v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const uint64_t i = i0 + (uint64_t)u * 256;
      v[u] = (MODE != 2 && i < hi) ? src[i] : make_ulonglong2(i, i);
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const uint64_t i = i0 + (uint64_t)u * 256;
      if (MODE == 1) acc += v[u].x ^ v[u].y;
      else if (i < hi) dst[i] = v[u];
    }
  }
  if (MODE == 1 && acc == 0x1234567ULL) sink[0] = acc;
}

__device__ __forceinline__ uint64_t ubench_mix(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// MODE 0: one 16-byte load of a random 16-byte record; 1: one agent-scope atomic add on its second word (what an
// occurrence costs the direct insert: the RMW of one random 64-byte sector); 2: load + atomic
template <int MODE, int PER>
__global__ __launch_bounds__(256) void k_ubench_rmw(uint64_t *tab, uint64_t nrec, uint64_t n, uint64_t seed, unsigned long long *sink)
{
  unsigned long long acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * PER) {
    uint64_t idx[PER];
#pragma unroll
    for (int p = 0; p < PER; p++) idx[p] = ubench_mix((i0 + p * stride) ^ seed) % nrec;
    if (MODE == 0 || MODE == 2) {
      ulonglong2 v[PER];
#pragma unroll
      for (int p = 0; p < PER; p++) v[p] = *reinterpret_cast<const ulonglong2 *>(tab + 2 * idx[p]);
#pragma unroll
      for (int p = 0; p < PER; p++) acc += v[p].x;
    }
    if (MODE >= 1) {
#pragma unroll
      for (int p = 0; p < PER; p++) __hip_atomic_fetch_add(tab + 2 * idx[p] + 1, 256ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 0x1234567ULL) sink[0] = acc;
}

template <class F> static int ubench_time(F launch, int reps, double *ms_out)
{
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess) return -1;
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return -1; }
  float ms = 0;
  bool ok = false;
  launch();
  if (hipDeviceSynchronize() == hipSuccess) {
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) launch();
    (void)hipEventRecord(e1, 0);
    ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (!ok || !(ms > 0.f) || reps < 1) { (void)hipGetLastError(); return -1; }  // (a rate is about to be divided by this)
  *ms_out = ms / reps;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace mcx
