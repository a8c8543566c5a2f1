// Microbenchmark (not product code): issue rules of a lone wavefront on gfx950 (which instruction pairs overlap).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 32768
#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                                                         \
  __global__ void name(unsigned *out, unsigned a, unsigned b) {                                                    \
    unsigned s0 = a, s1 = a + 1, s2 = a + 2, s3 = a + 3, s4 = a + 4, s5 = a + 5, s6 = a + 6, s7 = a + 7;            \
    unsigned v0 = a + threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7; \
    _Pragma("unroll 1") for (int i = 0; i < N; i++) asm volatile(REP8(body)                                         \
        : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7),                           \
          "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(b) : "scc", "vcc");  \
    out[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7; \
  }
// %0..%7 = s0..s7, %8..%15 = v0..v7, %16 = b
KERNEL(k_s_dep1, "s_add_u32 %0, %0, %16\n")
KERNEL(k_s_indep4, "s_add_u32 %0, %0, %16\n s_add_u32 %1, %1, %16\n s_add_u32 %2, %2, %16\n s_add_u32 %3, %3, %16\n")
KERNEL(k_s_noscc_dep, "s_lshl_b32 %0, %0, 1\n")            // writes scc too; compare with s_mov-like
KERNEL(k_s_mov_dep, "s_mov_b32 %1, %0\n s_mov_b32 %0, %1\n")
KERNEL(k_s_dep_fill1, "s_add_u32 %0, %0, %16\n s_xor_b32 %1, %1, %16\n")      // dep chain + 1 independent filler
KERNEL(k_s_dep_fill3, "s_add_u32 %0, %0, %16\n s_xor_b32 %1, %1, %16\n s_xor_b32 %2, %2, %16\n s_xor_b32 %3, %3, %16\n")
KERNEL(k_v_dep1, "v_add_u32 %8, %8, %9\n")
KERNEL(k_v_indep4, "v_add_u32 %8, %8, %12\n v_add_u32 %9, %9, %12\n v_add_u32 %10, %10, %12\n v_add_u32 %11, %11, %12\n")
KERNEL(k_v_s_alt, "v_add_u32 %8, %8, %9\n s_add_u32 %0, %0, %16\n")            // two dep chains, alternating units
KERNEL(k_v_s3, "v_add_u32 %8, %8, %9\n s_add_u32 %0, %0, %16\n s_add_u32 %1, %1, %16\n s_add_u32 %2, %2, %16\n")
KERNEL(k_v2_s2, "v_add_u32 %8, %8, %9\n v_add_u32 %10, %10, %9\n s_add_u32 %0, %0, %16\n s_add_u32 %1, %1, %16\n")
KERNEL(k_v_sdep, "v_add_u32 %8, %0, %8\n s_add_u32 %0, %0, %16\n")             // VALU reads SGPR just written
KERNEL(k_v_rfl, "v_add_u32 %8, %8, %9\n v_readfirstlane_b32 %0, %8\n")
KERNEL(k_rfl_s_v, "v_readfirstlane_b32 %0, %8\n s_add_u32 %0, %0, %16\n v_add_u32 %8, %0, %8\n")  // full round trip
KERNEL(k_rl_s_v, "v_readlane_b32 %0, %8, 5\n s_add_u32 %0, %0, %16\n v_add_u32 %8, %0, %8\n")
KERNEL(k_rl_nop_s_v, "v_readlane_b32 %0, %8, 5\n s_nop 0\n s_add_u32 %0, %0, %16\n v_add_u32 %8, %0, %8\n")
KERNEL(k_mulhi_mad, "v_mul_hi_u32 %9, %0, %8\n v_mul_hi_u32 %8, %0, %9\n")
KERNEL(k_dpp, "v_mov_b32_dpp %9, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32 %8, %9, %10\n")
KERNEL(k_v_ds, "v_add_u32 %8, %8, %9\n ds_write_b32 %10, %8\n")
KERNEL(k_v_nop0, "v_add_u32 %8, %8, %9\n s_nop 0\n")
KERNEL(k_v_nop1, "v_add_u32 %8, %8, %9\n s_nop 1\n")
KERNEL(k_v_nop3, "v_add_u32 %8, %8, %9\n s_nop 3\n")
KERNEL(k_v2_nop0, "v_add_u32 %8, %8, %9\n v_add_u32 %8, %8, %9\n s_nop 0\n")
__global__ void k_shl64(unsigned *out, unsigned a, unsigned b) {
  unsigned long long x = a + threadIdx.x;
  _Pragma("unroll 1") for (int i = 0; i < N; i++) asm volatile(REP8("v_lshlrev_b64 %0, 1, %0\n") : "+v"(x));
  out[blockIdx.x * 64 + threadIdx.x] = (unsigned)x;
}
__global__ void k_mad64(unsigned *out, unsigned a, unsigned b) {
  unsigned long long x = a + threadIdx.x; unsigned y = a * 3 + threadIdx.x;
  _Pragma("unroll 1") for (int i = 0; i < N; i++) asm volatile(REP8("v_mad_u64_u32 %0, vcc, %1, %1, %0\n") : "+v"(x) : "v"(y) : "vcc");
  out[blockIdx.x * 64 + threadIdx.x] = (unsigned)x;
}
KERNEL(k_mulhi1, "v_mul_hi_u32 %8, %8, %9\n")
KERNEL(k_ffbh, "v_ffbh_u32 %8, %8\n")
KERNEL(k_bitop3, "v_bitop3_b32 %8, %8, %9, %10 bitop3:0xf3\n")
KERNEL(k_add3, "v_add3_u32 %8, %8, %9, -1\n")
KERNEL(k_movdpp, "v_mov_b32_dpp %8, %9 wave_shr:1 row_mask:0x1 bank_mask:0x4\n")
KERNEL(k_adddpp, "v_add_u32_dpp %8, %9, %10 wave_shr:1 row_mask:0x1 bank_mask:0x4\n")
KERNEL(k_s_tail13,
       "s_sub_u32 %2, %6, %7\n s_add_u32 %0, %0, %7\n s_add_u32 %3, %0, %2\n s_xor_b32 %4, %0, %3\n s_flbit_i32_b32 %4, %4\n"
       "s_orn2_b32 %3, %3, %0\n s_lshl_b32 %3, %3, %4\n s_flbit_i32_b32 %3, %3\n s_add_u32 %4, %4, %3\n s_lshl_b32 %0, %0, %4\n"
       "s_lshl_b32 %1, %2, %4\n s_bitset1_b32 %1, 31\n s_bitset0_b32 %0, 31\n")
template <typename K> void run(const char *name, K k, int nblocks, unsigned *d, int ops_per_iter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(nblocks), dim3(64), 0, 0, d, 1u, 3u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(nblocks), dim3(64), 0, 0, d, 1u, 3u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ns = (ms * 1e6) / ((double)N * 8);
  printf("%-14s blocks=%5d  %8.3f ms  %7.3f ns/body  %6.3f ns/instr (%d instr)\n", name, nblocks, ms, ns, ns / ops_per_iter, ops_per_iter);
}
#define RUN(k, n) run(#k, k, nb, d, n)
int main() {
  setvbuf(stdout, 0, _IONBF, 0);
  unsigned *d; hipMalloc(&d, 1 << 24);
  for (int nb : {1}) {
    RUN(k_s_dep1, 1); RUN(k_s_indep4, 4); RUN(k_s_noscc_dep, 1); RUN(k_s_mov_dep, 2); RUN(k_s_dep_fill1, 2); RUN(k_s_dep_fill3, 4);
    RUN(k_v_dep1, 1); RUN(k_v_indep4, 4); RUN(k_v_s_alt, 2); RUN(k_v_s3, 4); RUN(k_v2_s2, 4); RUN(k_v_sdep, 2); RUN(k_v_rfl, 2);
    RUN(k_rfl_s_v, 3); RUN(k_rl_s_v, 3); RUN(k_rl_nop_s_v, 4); RUN(k_mulhi_mad, 2); RUN(k_dpp, 2); RUN(k_v_ds, 2); RUN(k_s_tail13, 13); RUN(k_v_nop0, 2); RUN(k_v_nop1, 2); RUN(k_v_nop3, 2); RUN(k_v2_nop0, 3); RUN(k_shl64, 1); RUN(k_mad64, 1); RUN(k_mulhi1, 1); RUN(k_ffbh, 1); RUN(k_bitop3, 1); RUN(k_add3, 1); RUN(k_movdpp, 1); RUN(k_adddpp, 1);
  }
  return 0;
}
