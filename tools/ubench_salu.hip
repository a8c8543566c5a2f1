// Microbenchmark (not product code): single-wave latency / issue cost of the scalar and vector integer ops an
// arithmetic-coder chain needs on gfx950, and of moving a value VALU -> SGPR -> VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 65536
#define REP8(x) x x x x x x x x
__global__ void k_sadd_dep(unsigned *out, unsigned a, unsigned b) {
  unsigned x = a;
#pragma unroll 1
  for (int i = 0; i < N; i++) asm volatile(REP8("s_add_u32 %0, %0, %1\n") : "+s"(x) : "s"(b) : "scc");
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
__global__ void k_sflbit_dep(unsigned *out, unsigned a, unsigned b) {
  unsigned x = a;
#pragma unroll 1
  for (int i = 0; i < N; i++) asm volatile(REP8("s_flbit_i32_b32 %0, %0\n") : "+s"(x) : "s"(b) : "scc");
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
__global__ void k_sshl_dep(unsigned *out, unsigned a, unsigned b) {
  unsigned x = a;
#pragma unroll 1
  for (int i = 0; i < N; i++) asm volatile(REP8("s_lshl_b32 %0, %0, %1\n") : "+s"(x) : "s"(b) : "scc");
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
__global__ void k_smulhi_dep(unsigned *out, unsigned a, unsigned b) {
  unsigned x = a;
#pragma unroll 1
  for (int i = 0; i < N; i++) asm volatile(REP8("s_mul_hi_u32 %0, %0, %1\n") : "+s"(x) : "s"(b));
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
__global__ void k_smul_indep(unsigned *out, unsigned a, unsigned b) {  // 8 independent s_mul per iteration
  unsigned x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
#pragma unroll 1
  for (int i = 0; i < N; i++)
    asm volatile("s_mul_hi_u32 %0, %0, %8\n s_mul_i32 %1, %1, %8\n s_mul_hi_u32 %2, %2, %8\n s_mul_i32 %3, %3, %8\n"
                 "s_mul_hi_u32 %4, %4, %8\n s_mul_i32 %5, %5, %8\n s_mul_hi_u32 %6, %6, %8\n s_mul_i32 %7, %7, %8\n"
                 : "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3), "+s"(x4), "+s"(x5), "+s"(x6), "+s"(x7) : "s"(b));
  if (threadIdx.x == 0) out[blockIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_vadd_dep(unsigned *out, unsigned a, unsigned b) {
  unsigned x = a + threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < N; i++) asm volatile(REP8("v_add_u32 %0, %0, %1\n") : "+v"(x) : "v"(b));
  out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void k_vadd_indep(unsigned *out, unsigned a, unsigned b) {  // 8 independent chains
  unsigned x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
#pragma unroll 1
  for (int i = 0; i < N; i++)
    asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                 "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_vadd_indep2(unsigned *out, unsigned a, unsigned b) {  // 2 independent chains
  unsigned x0 = a, x1 = a + 1;
#pragma unroll 1
  for (int i = 0; i < N; i++)
    asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n"
                 "v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n"
                 : "+v"(x0), "+v"(x1) : "v"(b));
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1;
}
__global__ void k_vmulhi_indep(unsigned *out, unsigned a, unsigned b) {
  unsigned x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
#pragma unroll 1
  for (int i = 0; i < N; i++)
    asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                 "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// VALU -> SGPR -> VALU round trip: v_add (sgpr operand) ; v_readfirstlane
__global__ void k_roundtrip(unsigned *out, unsigned a, unsigned b) {
  unsigned v = a + threadIdx.x, s = a;
#pragma unroll 1
  for (int i = 0; i < N; i++)
    asm volatile(REP8("v_add_u32 %0, %1, %0\n v_readfirstlane_b32 %1, %0\n") : "+v"(v), "+s"(s));
  out[blockIdx.x * 64 + threadIdx.x] = v + s;
}
// v_readlane with constant lane + one SALU op + VALU use
__global__ void k_roundtrip2(unsigned *out, unsigned a, unsigned b) {
  unsigned v = a + threadIdx.x, s = a;
#pragma unroll 1
  for (int i = 0; i < N; i++)
    asm volatile(REP8("v_mul_hi_u32 %0, %1, %0\n v_readlane_b32 %1, %0, 5\n s_add_u32 %1, %1, %2\n") : "+v"(v), "+s"(s) : "s"(b) : "scc");
  out[blockIdx.x * 64 + threadIdx.x] = v + s;
}
// the hybrid chain shape: mul_hi -> mad64 -> readlane x2 -> ~12 SALU -> repeat
__global__ void k_hybrid(unsigned *out, unsigned a, unsigned b) {
  unsigned gz = a * 2654435761u + threadIdx.x, gw = 0x7fffffffu - threadIdx.x * 977u;
  unsigned sM = 0xC0000000u, slo = 0x1234567u, sA, sB, t0, t1, t2;
  unsigned acc;
  unsigned vt;
#pragma unroll 1
  for (int i = 0; i < N; i++)
    asm volatile(REP8(
        "v_mul_hi_u32 %[vt], %[sM], %[gz]\n"
        "v_mul_hi_u32 %[acc], %[gw], %[vt]\n"     // stands for v_mad_u64_u32 (same latency class)
        "v_readlane_b32 %[sA], %[acc], 7\n"
        "v_readlane_b32 %[sB], %[acc], 39\n"
        "s_sub_u32 %[t0], %[sA], %[sB]\n"          // W
        "s_add_u32 %[slo], %[slo], %[sB]\n"        // nlo
        "s_add_u32 %[t1], %[slo], %[t0]\n"         // nhi+1
        "s_xor_b32 %[t2], %[slo], %[t1]\n"         // x
        "s_flbit_i32_b32 %[t2], %[t2]\n"           // k
        "s_orn2_b32 %[t1], %[t1], %[slo]\n"
        "s_lshl_b32 %[t1], %[t1], %[t2]\n"
        "s_flbit_i32_b32 %[t1], %[t1]\n"           // u
        "s_add_u32 %[t2], %[t2], %[t1]\n"
        "s_lshl_b32 %[slo], %[slo], %[t2]\n"
        "s_lshl_b32 %[sM], %[t0], %[t2]\n"
        "s_bitset1_b32 %[sM], 31\n"
        "s_bitset0_b32 %[slo], 31\n")
        : [vt] "=&v"(vt), [acc] "=&v"(acc), [sM] "+s"(sM), [slo] "+s"(slo), [sA] "=&s"(sA), [sB] "=&s"(sB), [t0] "=&s"(t0),
          [t1] "=&s"(t1), [t2] "=&s"(t2)
        : [gz] "v"(gz), [gw] "v"(gw) : "vcc", "scc");
  out[blockIdx.x * 64 + threadIdx.x] = sM + slo;
}
// all-scalar chain shape: 6 s_mul (2 x {mul_hi, mul_lo, mul_hi}) + add/addc + same SALU tail; operands by 4 v_readlane
__global__ void k_allscalar(unsigned *out, unsigned a, unsigned b) {
  unsigned vg0 = a * 2654435761u + threadIdx.x, vg1 = 0x7fffffffu - threadIdx.x * 977u, vg2 = vg0 * 3, vg3 = vg1 - 12345;
  unsigned sM = 0xC0000000u, slo = 0x1234567u, sA, sB, t0, t1, t2, g0, g1, g2, g3, m0, m1, m2, m3;
#pragma unroll 1
  for (int i = 0; i < N; i++)
    asm volatile(REP8(
        "v_readlane_b32 %[g0], %[vg0], 7\n"
        "v_readlane_b32 %[g1], %[vg1], 7\n"
        "v_readlane_b32 %[g2], %[vg2], 7\n"
        "v_readlane_b32 %[g3], %[vg3], 7\n"
        "s_mul_hi_u32 %[m0], %[sM], %[g0]\n"
        "s_mul_i32 %[m1], %[sM], %[g1]\n"
        "s_mul_hi_u32 %[sA], %[sM], %[g1]\n"
        "s_mul_hi_u32 %[m2], %[sM], %[g2]\n"
        "s_mul_i32 %[m3], %[sM], %[g3]\n"
        "s_mul_hi_u32 %[sB], %[sM], %[g3]\n"
        "s_add_u32 %[m0], %[m0], %[m1]\n"
        "s_addc_u32 %[sA], %[sA], 0\n"
        "s_add_u32 %[m2], %[m2], %[m3]\n"
        "s_addc_u32 %[sB], %[sB], 0\n"
        "s_sub_u32 %[t0], %[sA], %[sB]\n"          // W
        "s_add_u32 %[slo], %[slo], %[sB]\n"        // nlo
        "s_add_u32 %[t1], %[slo], %[t0]\n"         // nhi+1
        "s_xor_b32 %[t2], %[slo], %[t1]\n"         // x
        "s_flbit_i32_b32 %[t2], %[t2]\n"           // k
        "s_orn2_b32 %[t1], %[t1], %[slo]\n"
        "s_lshl_b32 %[t1], %[t1], %[t2]\n"
        "s_flbit_i32_b32 %[t1], %[t1]\n"           // u
        "s_add_u32 %[t2], %[t2], %[t1]\n"
        "s_lshl_b32 %[slo], %[slo], %[t2]\n"
        "s_lshl_b32 %[sM], %[t0], %[t2]\n"
        "s_bitset1_b32 %[sM], 31\n"
        "s_bitset0_b32 %[slo], 31\n")
        : [sM] "+s"(sM), [slo] "+s"(slo), [sA] "=&s"(sA), [sB] "=&s"(sB), [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2),
          [g0] "=&s"(g0), [g1] "=&s"(g1), [g2] "=&s"(g2), [g3] "=&s"(g3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
        : [vg0] "v"(vg0), [vg1] "v"(vg1), [vg2] "v"(vg2), [vg3] "v"(vg3) : "scc");
  out[blockIdx.x * 64 + threadIdx.x] = sM + slo;
}
// v_writelane cost (independent)
__global__ void k_writelane(unsigned *out, unsigned a, unsigned b) {
  unsigned v = a + threadIdx.x, s = a;
#pragma unroll 1
  for (int i = 0; i < N; i++)
    asm volatile(REP8("s_add_u32 %1, %1, 1\n v_writelane_b32 %0, %1, 9\n") : "+v"(v), "+s"(s) : : "scc");
  out[blockIdx.x * 64 + threadIdx.x] = v + s;
}
template <typename K> void run(const char *name, K k, int nblocks, unsigned *d, int ops_per_iter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(nblocks), dim3(64), 0, 0, d, 1u, 3u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(nblocks), dim3(64), 0, 0, d, 1u, 3u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ns_per_op = (ms * 1e6) / ((double)N * ops_per_iter);
  printf("%-16s blocks=%5d  %8.3f ms  %7.3f ns/op  (%6.2f cycles @2.4GHz)\n", name, nblocks, ms, ns_per_op, ns_per_op * 2.4);
}
int main() {
  setvbuf(stdout, 0, _IONBF, 0);
  unsigned *d; hipMalloc(&d, 1 << 24);
  for (int nb : {1, 480}) {
    run("sadd_dep", k_sadd_dep, nb, d, 8);
    run("sflbit_dep", k_sflbit_dep, nb, d, 8);
    run("sshl_dep", k_sshl_dep, nb, d, 8);
    run("smulhi_dep", k_smulhi_dep, nb, d, 8);
    run("smul_indep", k_smul_indep, nb, d, 8);
    run("vadd_dep", k_vadd_dep, nb, d, 8);
    run("vadd_indep8", k_vadd_indep, nb, d, 8);
    run("vadd_indep2", k_vadd_indep2, nb, d, 8);
    run("vmulhi_indep8", k_vmulhi_indep, nb, d, 8);
    run("roundtrip(2op)", k_roundtrip, nb, d, 8);
    run("roundtrip2(3op)", k_roundtrip2, nb, d, 8);
    run("hybrid(step)", k_hybrid, nb, d, 8);
    run("allscalar(step)", k_allscalar, nb, d, 8);
    run("writelane(2op)", k_writelane, nb, d, 8);
  }
  return 0;
}
