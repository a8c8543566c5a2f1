// Microbenchmark (not product code): how fast does ONE wavefront issue dependent / independent
// scalar and vector integer ops on gfx950?  Informs the arithmetic-coder chain design.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void k_salu_dep(unsigned *out, unsigned a, unsigned b) {
  unsigned x = a;
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    asm volatile("s_add_u32 %0, %0, %1\n s_xor_b32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_xor_b32 %0, %0, %1\n"
                 "s_add_u32 %0, %0, %1\n s_xor_b32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_xor_b32 %0, %0, %1\n" : "+s"(x) : "s"(b));
  }
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
__global__ void k_salu_indep(unsigned *out, unsigned a, unsigned b) {
  unsigned x = a, y = a + 1, z = a + 2, w = a + 3;
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    asm volatile("s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4\n"
                 "s_xor_b32 %0, %0, %4\n s_xor_b32 %1, %1, %4\n s_xor_b32 %2, %2, %4\n s_xor_b32 %3, %3, %4\n"
                 : "+s"(x), "+s"(y), "+s"(z), "+s"(w) : "s"(b));
  }
  if (threadIdx.x == 0) out[blockIdx.x] = x + y + z + w;
}
__global__ void k_smul_dep(unsigned *out, unsigned a, unsigned b) {
  unsigned x = a;
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    asm volatile("s_mul_hi_u32 %0, %0, %1\n s_mul_i32 %0, %0, %1\n s_mul_hi_u32 %0, %0, %1\n s_mul_i32 %0, %0, %1\n"
                 "s_mul_hi_u32 %0, %0, %1\n s_mul_i32 %0, %0, %1\n s_mul_hi_u32 %0, %0, %1\n s_mul_i32 %0, %0, %1\n" : "+s"(x) : "s"(b));
  }
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
__global__ void k_valu_dep(unsigned *out, unsigned a, unsigned b) {
  unsigned x = a + threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
                 "v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n" : "+v"(x) : "v"(b));
  }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void k_vmad64_dep(unsigned *out, unsigned a, unsigned b) {
  unsigned long long x = a + threadIdx.x;
  unsigned bb = b;
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0\n v_mad_u64_u32 %0, vcc, %1, %1, %0\n v_mad_u64_u32 %0, vcc, %1, %1, %0\n v_mad_u64_u32 %0, vcc, %1, %1, %0\n"
                 "v_mad_u64_u32 %0, vcc, %1, %1, %0\n v_mad_u64_u32 %0, vcc, %1, %1, %0\n v_mad_u64_u32 %0, vcc, %1, %1, %0\n v_mad_u64_u32 %0, vcc, %1, %1, %0\n"
                 : "+v"(x) : "v"(bb) : "vcc");
  }
  out[blockIdx.x * 64 + threadIdx.x] = (unsigned)x;
}
__global__ void k_vmulhi_dep(unsigned *out, unsigned a, unsigned b) {
  unsigned x = a + threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_hi_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n"
                 "v_mul_hi_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n v_mul_hi_u32 %0, %0, %1\n v_mul_lo_u32 %0, %0, %1\n" : "+v"(x) : "v"(b));
  }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void k_mixed(unsigned *out, unsigned a, unsigned b) {  // alternate SALU / VALU, independent
  unsigned x = a, v = a + threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    asm volatile("s_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %3\n s_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %3\n"
                 "s_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %3\n s_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %3\n" : "+s"(x), "+v"(v) : "s"(b), "v"(b));
  }
  out[blockIdx.x * 64 + threadIdx.x] = x + v;
}
__global__ void k_readlane(unsigned *out, unsigned a, unsigned b) {
  unsigned v = a + threadIdx.x, x = 0;
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    unsigned t0, t1, t2, t3;
    asm volatile("v_readlane_b32 %0, %5, %4\n v_readlane_b32 %1, %5, %4\n v_readlane_b32 %2, %5, %4\n v_readlane_b32 %3, %5, %4\n"
                 : "=s"(t0), "=s"(t1), "=s"(t2), "=s"(t3) : "s"(i & 63), "v"(v));
    asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %2\n s_add_u32 %0, %0, %3\n s_add_u32 %0, %0, %4\n" : "+s"(x) : "s"(t0), "s"(t1), "s"(t2), "s"(t3));
  }
  if (threadIdx.x == 0) out[blockIdx.x] = x;
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
  printf("%-14s blocks=%5d  %.3f ms  %.3f ns/op  (%.2f cycles @2.4GHz)\n", name, nblocks, ms, ns_per_op, ns_per_op * 2.4);
}
int main() {
  setvbuf(stdout, 0, _IONBF, 0);
  unsigned *d; hipMalloc(&d, 1 << 24);
  for (int nb : {1, 1024, 4096}) {
    run("salu_dep", k_salu_dep, nb, d, 8);
    run("salu_indep", k_salu_indep, nb, d, 8);
    run("smul_dep", k_smul_dep, nb, d, 8);
    run("valu_dep", k_valu_dep, nb, d, 8);
    run("vmad64_dep", k_vmad64_dep, nb, d, 8);
    run("vmulhi_dep", k_vmulhi_dep, nb, d, 8);
    run("mixed", k_mixed, nb, d, 8);
    run("readlane4+4", k_readlane, nb, d, 8);
  }
  return 0;
}
