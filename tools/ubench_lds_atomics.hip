// LDS atomic throughput on gfx950: returning vs non-returning 32-bit adds on pseudo-random words (the trigram counters'
// access pattern), 512-thread workgroups, one per CU.   hipcc --offload-arch=gfx950 -O3 -o ubench_lds_atomics ubench_lds_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int WORDS = 30500;
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned *out, unsigned iters, unsigned span) {
  __shared__ unsigned tab[WORDS];
  for (int i = threadIdx.x; i < WORDS; i += 512) tab[i] = 0;
  __syncthreads();
  unsigned x = blockIdx.x * 512 + threadIdx.x + 12345, acc = 0;
  for (unsigned i = 0; i < iters; i++) {
    x = x * 1664525u + 1013904223u;
    const unsigned idx = (x >> 8) % span;
    if (MODE == 0) acc += atomicAdd(&tab[idx], 1u);            // returning
    else if (MODE == 1) atomicAdd(&tab[idx], 1u);              // result unused: ds_add_u32
    else { const unsigned old = atomicAdd(&tab[idx >> 1], 1u << ((idx & 1) * 16)); if (((old >> ((idx & 1) * 16)) & 0xFFFFu) == 0x7FFFu) acc++; }  // the trigram kernel's form
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc + tab[1];
}
int main() {
  unsigned *d;
  hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned iters = 40000;
  for (unsigned span : {30500u, 3000u, 64u}) {
    for (int mode = 0; mode < 3; mode++) {
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, 256, 512, 0, 0, d, iters, span);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, 256, 512, 0, 0, d, iters, span);
        else hipLaunchKernelGGL(k<2>, 256, 512, 0, 0, d, iters, span);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double n = 256.0 * 512 * iters;
      printf("span %6u mode %d (%s): %.2f ms, %.2f G atomics/s, %.2f per CU per cycle at 2.4 GHz\n", span, mode,
             mode == 0 ? "returning" : mode == 1 ? "non-returning" : "16-bit fields, returning", ms, n / ms / 1e6, n / ms / 1e6 / 256 / 2.4);
    }
  }
  return 0;
}
