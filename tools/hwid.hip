// Diagnostic (not product code): where do the waves of a 2-wave workgroup land?  Records HW_ID / XCC_ID per wave
// for a grid shaped like ac_encode_k's (one 128-thread workgroup per block), kept resident by a spin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(128) void k(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned x = threadIdx.x;
  for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (x == 12345u);
  }
}
int main(int argc, char **argv) {
  int nb = argc > 1 ? atoi(argv[1]) : 477;
  int nk = argc > 2 ? atoi(argv[2]) : 1;
  std::vector<unsigned *> d(nk);
  std::vector<hipStream_t> st(nk);
  for (int i = 0; i < nk; i++) { hipMalloc(&d[i], nb * 4 * sizeof(unsigned)); hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); }
  for (int i = 0; i < nk; i++) hipLaunchKernelGGL(k, dim3(nb), dim3(128), 0, st[i], d[i], 2000000);
  hipDeviceSynchronize();
  std::map<unsigned, std::vector<int>> percu;  // (xcc, se, cu) -> list of simd of wave0 / wave1
  std::map<unsigned, int> chain_simd;         // (xcc,se,cu,simd) -> number of wave0s
  int hist_simd[2][4] = {{0}};
  for (int i = 0; i < nk; i++) {
    std::vector<unsigned> h(nb * 4);
    hipMemcpy(h.data(), d[i], nb * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int b = 0; b < nb; b++)
      for (int w = 0; w < 2; w++) {
        unsigned hw = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1] & 0xF;
        unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        percu[key].push_back(w * 4 + simd);
        hist_simd[w][simd]++;
        if (w == 0) chain_simd[(key << 2) | simd]++;
        if (b < 6 && i == 0) printf("kernel %d block %d wave %d: hw=%08x xcc=%u se=%u sh=%u cu=%u simd=%u waveslot=%u\n", i, b, w, hw, xcc, se, sh, cu, simd, hw & 15);
      }
  }
  printf("CUs used: %zu\n", percu.size());
  std::map<size_t, int> wgs_per_cu;
  for (auto &kv : percu) wgs_per_cu[kv.second.size() / 2]++;
  for (auto &kv : wgs_per_cu) printf("  %zu workgroups on a CU: %d CUs\n", kv.first, kv.second);
  printf("SIMD of wave 0: %d %d %d %d   wave 1: %d %d %d %d\n", hist_simd[0][0], hist_simd[0][1], hist_simd[0][2], hist_simd[0][3],
         hist_simd[1][0], hist_simd[1][1], hist_simd[1][2], hist_simd[1][3]);
  std::map<int, int> share;
  for (auto &kv : chain_simd) share[kv.second]++;
  for (auto &kv : share) printf("  SIMDs holding %d wave-0s: %d\n", kv.first, kv.second);
  return 0;
}
