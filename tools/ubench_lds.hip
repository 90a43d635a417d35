// Microbenchmark: cost of the LDS operations of the LDS insert (mcx_defer.h) at random addresses,
// 2 workgroups x 512 threads x 64 KiB per CU as in k_lds_insert.  Reports LDS-pipe cycles per
// wave-instruction (CU cycles at 2.4 GHz / wave-instructions issued per CU).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o ubench_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define LDS_AS __attribute__((address_space(3)))
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// MODE 0: 2 x ds_read_b128 of a random 64-byte bucket (first 32 bytes)
//      1: ds_add_rtn_u64 random slot      2: ds_add_u64 (no return)
//      3: ds_add_rtn_u32 random slot      4: ds_add_u32 (no return)
//      5: ds_or_b64 on ~1/16 of the lanes 6: ds_read_b64 random slot
//      7: 2 x ds_read_b128 with the half-swizzled 32-byte key layout (16 positions per read)
//      8: ds_cmpst_rtn_b64 random slot (always failing compare)
//      9: mode 0 + mode 1 (dependent: the add goes to the bucket that was read)
template <int MODE>
__global__ __launch_bounds__(512, 4) void k(u64 *sink, int iters)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  u64 *lds = reinterpret_cast<u64 *>(dyn);
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
  __syncthreads();
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
  u64 acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t r = lcg(s);
      const uint32_t b = r & 1023u;   // bucket
      const uint32_t sl = r & 4095u;  // slot
      if (MODE == 0 || MODE == 9) {
        const u64x2 a = *(LDS_AS const volatile u64x2 *)(lds + b * 8);
        const u64x2 c = *(LDS_AS const volatile u64x2 *)(lds + b * 8 + 2);
        acc += a.x ^ a.y ^ c.x ^ c.y;
      }
      if (MODE == 7) {
        const uint32_t sw = (b >> 3) & 1u;
        const u64x2 a = *(LDS_AS const volatile u64x2 *)(lds + b * 4 + 2 * sw);
        const u64x2 c = *(LDS_AS const volatile u64x2 *)(lds + b * 4 + 2 * (sw ^ 1u));
        acc += a.x ^ a.y ^ c.x ^ c.y;
      }
      if (MODE == 1) acc += atomicAdd(lds + 4096 + sl, 256ULL);
      if (MODE == 9) acc += atomicAdd(lds + b * 8 + 4 + (acc & 3), 256ULL);
      if (MODE == 2) atomicAdd(lds + 4096 + sl, 256ULL);
      if (MODE == 3) acc += atomicAdd(reinterpret_cast<unsigned *>(lds) + 8192 + sl, 256u);
      if (MODE == 4) atomicAdd(reinterpret_cast<unsigned *>(lds) + 8192 + sl, 256u);
      if (MODE == 5) { if ((r >> 12 & 15u) == 0) atomicOr(lds + 4096 + sl, 5ULL); }
      if (MODE == 6) acc += *(LDS_AS const volatile u64 *)(lds + sl);
      if (MODE == 8) acc += atomicCAS(lds + 4096 + sl, ~0ULL, 1ULL);
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

template <int MODE> static void run(const char *name, u64 *sink)
{
  const int iters = 2000, blocks = 512;  // 2 per CU on 256 CUs
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 512, 65536>>>(sink, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE><<<blocks, 512, 65536>>>(sink, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double winstr_per_cu = 2.0 * 8 /*waves*/ * iters * 8;  // wave-iterations per CU
  printf("%-44s %8.3f ms  %7.1f CU-cycles per wave-iteration  (%.2f G lane-ops/s)\n", name, ms,
         ms * 1e-3 * 2.4e9 / winstr_per_cu, 512.0 * blocks * iters * 8 / (ms * 1e-3) / 1e9);
}

int main()
{
  u64 *sink;
  hipMalloc(&sink, 8);
  run<0>("2 x ds_read_b128, 64-B buckets (current)", sink);
  run<7>("2 x ds_read_b128, 32-B keys half-swizzled", sink);
  run<6>("ds_read_b64 random slot", sink);
  run<1>("ds_add_rtn_u64 random slot", sink);
  run<2>("ds_add_u64 random slot", sink);
  run<3>("ds_add_rtn_u32 random slot", sink);
  run<4>("ds_add_u32 random slot", sink);
  run<5>("ds_or_b64 on 1/16 of the lanes", sink);
  run<8>("ds_cmpst_rtn_b64 random slot", sink);
  run<9>("bucket read + dependent ds_add_rtn_u64", sink);
  return 0;
}
