// Microbenchmark: issue cost of the integer VALU instructions the partition kernels are made of
// (gfx950, wave64 on a 16-lane SIMD: a full-rate instruction occupies the SIMD for 4 cycles).
// 4 waves per SIMD, 8 independent dependency chains per lane, so the figure is the issue rate and
// not the latency.  Reports SIMD cycles per wave-instruction at 2.4 GHz.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { ADD, XOR, MUL_LO, MUL_HI, MUL_U24, MAD_U24, MAD_U64, SHL64, SHR64, CMP64, ALIGNBIT, BFE, PERM, LSHL_ADD, LSHL_OR, AND_OR,
       BFREV, ADD64, CNDMASK, MUL_LO_I, XAD, ADD3, NMODES };
static const char *names[NMODES] = {"v_add_u32", "v_xor_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24",
                                    "v_mad_u64_u32", "v_lshlrev_b64", "v_lshrrev_b64", "v_cmp_lt_u64 + v_cndmask", "v_alignbit_b32",
                                    "v_bfe_u32", "v_perm_b32", "v_lshl_add_u32", "v_lshl_or_b32", "v_and_or_b32", "v_bfrev_b32",
                                    "v_add_co + v_addc (64-bit add)", "v_cmp_lt_u32 + v_cndmask", "v_mul_i32_i24", "v_xad_u32", "v_add3_u32"};

template <int MODE> __global__ __launch_bounds__(256, 4) void k(uint32_t *sink, int iters, uint32_t seed)
{
  uint32_t a[8], b[8];
  uint64_t w[8];
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 2654435761u + seed + i; b[i] = a[i] ^ 0x9E3779B9u; w[i] = ((uint64_t)a[i] << 32) | b[i]; }
  const uint32_t c = seed | 1u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#define ONE(i)                                                                                                         \
  if (MODE == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                          \
  if (MODE == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                          \
  if (MODE == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                    \
  if (MODE == MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                    \
  if (MODE == MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                  \
  if (MODE == MUL_LO_I) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                 \
  if (MODE == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));                   \
  if (MODE == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(c) : "vcc");      \
  if (MODE == SHL64) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(w[i]));                                              \
  if (MODE == SHR64) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(w[i]) : "v"(c & 7u));                               \
  if (MODE == CMP64) asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]), "v"(c) : "vcc"); \
  if (MODE == CNDMASK) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]), "v"(c) : "vcc"); \
  if (MODE == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b[i]));                          \
  if (MODE == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a[i]));                                                \
  if (MODE == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                         \
  if (MODE == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(c));                             \
  if (MODE == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(c));                               \
  if (MODE == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));                     \
  if (MODE == BFREV) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i]));                                                   \
  if (MODE == XAD) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));                           \
  if (MODE == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));                         \
  if (MODE == ADD64) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(c), "v"(seed) : "vcc");
      REP8(ONE)
    }
  }
  uint32_t acc = 0;
  for (int i = 0; i < 8; i++) acc ^= a[i] ^ b[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  if (acc == 0x12345671u) sink[0] = acc;
}

template <int MODE> static void run(uint32_t *sink)
{
  const int iters = 4000, blocks = 1024;  // 4 blocks x 4 waves per CU = 4 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(sink, 10, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(sink, iters, 12345u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int per_stmt = (MODE == CMP64 || MODE == CNDMASK || MODE == ADD64) ? 2 : 1;
  const double stmts_per_simd = 4.0 /*waves*/ * iters * 64.0;
  printf("%-34s %8.3f ms  %6.2f SIMD cycles per statement (%d instruction%s)\n", names[MODE], ms,
         ms * 1e-3 * 2.4e9 / stmts_per_simd, per_stmt, per_stmt > 1 ? "s" : "");
}

template <int M> struct Runner { static void go(uint32_t *s) { run<M>(s); Runner<M + 1>::go(s); } };
template <> struct Runner<NMODES> { static void go(uint32_t *) {} };

int main()
{
  uint32_t *sink;
  hipMalloc(&sink, 64);
  Runner<0>::go(sink);
  return 0;
}
