// Microbenchmark (not product code): what LDS and vector-memory instructions cost a lone wavefront on gfx950, next to
// four VALU instructions -- the budget question behind ac_encode_lanes_k (kernels_acl.hpp).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define N 16384
#define REP8(x) x x x x x x x x
#define V4 "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
// %0..%3 VALU regs, %4 constant, %5 lds address (lane * 16), %6:%7.. data regs, %8 = 64-bit global address (per lane), %9 sgpr pair
#define KERNEL(name, body)                                                                                             \
  __global__ void name(unsigned *out, unsigned a) {                                                                    \
    __shared__ uint4 lds[4096];                                                                                        \
    unsigned v0 = a + threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, c = a;                                        \
    unsigned la = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 16384;                                                 \
    u32x4 d = {v0, v1, v2, v3};                                                                              \
    unsigned *gp = out + 4096 + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4096;                                \
    unsigned long long sv = 0, d2 = a;                                                                                        \
    lds[threadIdx.x] = make_uint4(d.x, d.y, d.z, d.w);                                                                                              \
    __syncthreads();                                                                                                   \
    _Pragma("unroll 1") for (int i = 0; i < N; i++) {                                                                  \
      asm volatile(REP8(body) "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                        \
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(c), "+v"(la), "+v"(d), "+v"(gp), "+s"(sv), "+v"(d2)::"vcc", "scc", "memory"); \
    }                                                                                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + d.x + d.y + d.z + d.w;                             \
  }
KERNEL(k_v4, V4)
KERNEL(k_v8, V4 V4)
KERNEL(k_v4_dsr128, V4 "ds_read_b128 %6, %5\n")
KERNEL(k_v4_dsr64, V4 "ds_read_b64 %9, %5\n")
KERNEL(k_v4_dsw64, V4 "ds_write_b64 %5, %9\n")
KERNEL(k_v4_dsw128, V4 "ds_write_b128 %5, %6\n")
KERNEL(k_v8_dsr128_dsw64, V4 "ds_read_b128 %6, %5\n" V4 "ds_write_b64 %5, %9 offset:8192\n")
KERNEL(k_v4_gst, V4 "global_store_dword %7, %0, off\n")
KERNEL(k_v4_gst_m, V4 "s_mov_b64 exec, 0x01010101\n global_store_dword %7, %0, off\n s_mov_b64 exec, -1\n")
KERNEL(k_v4_saveexec, V4 "v_cmp_lt_u32 vcc, 31, %0\n s_and_saveexec_b64 %8, vcc\n s_cbranch_execz 1f\n v_mov_b32 %1, %2\n1:\n s_or_b64 exec, exec, %8\n")
KERNEL(k_v4_addc, V4 "v_add_co_u32 %0, vcc, %0, %4\n s_nop 1\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n s_nop 1\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n")
KERNEL(k_v4_addc_nonop, V4 "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n")
KERNEL(k_v4_gldlds, V4 "s_mov_b32 m0, 0\n global_load_lds_dwordx4 %7, off\n")
KERNEL(k_v4_gld128, V4 "global_load_dwordx4 %6, %7, off\n")
KERNEL(k_v4_vccbr, V4 "v_cmp_lt_u32 vcc, %0, %0\n s_cbranch_vccnz 1f\n1:\n")
template <typename K> void run(const char *name, K k, int nthreads, unsigned *d, int ninstr) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(1), dim3(nthreads), 0, 0, d, 1u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ns = (ms * 1e6) / ((double)N * 8);
  printf("%-20s waves=%d  %7.2f ns/body  = v4 + %6.2f ns   (%d instr in the body)\n", name, nthreads / 64, ns, ns - 7.0, ninstr);
}
#define RUN(k, n) run(#k, k, nt, d, n)
int main() {
  setvbuf(stdout, 0, _IONBF, 0);
  unsigned *d; hipMalloc(&d, 1ull << 30);
  for (int nt : {64, 192}) {
    RUN(k_v4, 4); RUN(k_v8, 8); RUN(k_v4_dsr128, 5); RUN(k_v4_dsr64, 5); RUN(k_v4_dsw64, 5); RUN(k_v4_dsw128, 5); RUN(k_v8_dsr128_dsw64, 10);
    RUN(k_v4_gst, 5); RUN(k_v4_gst_m, 7); RUN(k_v4_saveexec, 9); RUN(k_v4_addc, 9); RUN(k_v4_addc_nonop, 7); RUN(k_v4_gldlds, 6); RUN(k_v4_gld128, 5); RUN(k_v4_vccbr, 6);
  }
  return 0;
}
