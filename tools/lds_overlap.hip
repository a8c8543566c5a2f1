// Diagnostic (not product code): what in an ac_encode_k-shaped resident grid slows an LDS-atomic kernel down?
#include <hip/hip_runtime.h>
#include <cstdio>
// mode 0: plain VALU spin; 1: DPP wave_shr chain; 2: DPP row_shr chain (stays inside 16 lanes); 3: ds_swizzle-free LDS traffic
__global__ __launch_bounds__(128) void resident(unsigned *out, int spin, int mode) {
  __shared__ unsigned lds[660];
  unsigned x = threadIdx.x, y = x * 3;
  lds[threadIdx.x] = x;
  if (mode == 0) for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
  if (mode == 1) for (int i = 0; i < spin; i++) { y = __builtin_amdgcn_update_dpp(y, x, 0x138, 0xF, 0xF, false); x = y + 12345u; x ^= y >> 3; }
  if (mode == 2) for (int i = 0; i < spin; i++) { y = __builtin_amdgcn_update_dpp(y, x, 0x111, 0xF, 0xF, false); x = y + 12345u; x ^= y >> 3; }
  if (mode == 3) for (int i = 0; i < spin; i++) { lds[(x >> 7) & 511] += x; x = x * 1664525u + lds[threadIdx.x]; }
  if (x == 12345u) out[blockIdx.x] = x + lds[5] + y;
}
__global__ __launch_bounds__(1024) void needy(unsigned *out, int spin) {
  extern __shared__ unsigned dyn[];
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
  for (int i = threadIdx.x; i < 32000; i += 1024) dyn[i] = 0;
  __syncthreads();
  for (int i = 0; i < spin; i++) { x = x * 1664525u + 1013904223u; atomicAdd(&dyn[(x >> 12) % 3000u], 1u); }
  __syncthreads();
  if (x == 12345u) out[blockIdx.x] = x + dyn[7];
}
int main() {
  unsigned *d; hipMalloc(&d, 1 << 20);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void *)needy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int mode = -1; mode < 4; mode++) {
    if (mode >= 0) hipLaunchKernelGGL(resident, dim3(477), dim3(128), 0, a, d, 30000000, mode);
    hipEventRecord(e0, b);
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(needy, dim3(256), dim3(1024), 128000, b, d, 2000);
    hipEventRecord(e1, b);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("resident mode %2d: LDS-atomic kernel %.3f ms per launch\n", mode, ms / 4);
    hipDeviceSynchronize();
  }
  return 0;
}
