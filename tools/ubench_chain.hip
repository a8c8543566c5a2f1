// Microbenchmark (not product code): ns per symbol of candidate arithmetic-coder state chains on gfx950.
// One wavefront per block runs `iters` rounds of 64 symbols with operands held in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32; typedef unsigned long long u64;

__device__ __forceinline__ u32 mulfrac_a(u32 R, u32 g_lo, u32 g_hi) {
  const u64 t0 = (u64)R * g_lo + g_lo;
  const u64 t1 = (u64)R * g_hi + g_hi + (t0 >> 32);
  return (u32)(t1 >> 32);
}
// explicit 32-bit formulation (no 33-bit R+1)
__device__ __forceinline__ u32 mulfrac_b(u32 R, u32 g_lo, u32 g_hi) {
  u32 h0 = __umulhi(R, g_lo), l0 = R * g_lo;
  u32 s0 = l0 + g_lo; h0 += (s0 < l0);
  u32 h1 = __umulhi(R, g_hi), l1 = R * g_hi;
  u32 s1 = l1 + g_hi; h1 += (s1 < l1);
  u32 s2 = s1 + h0; h1 += (s2 < s1);
  return h1;
}

template <int V> __device__ __forceinline__ void step(u32 &lo, u32 &hi, u32 glo0, u32 glo1, u32 ghi0, u32 ghi1, u32 &k_out, u32 &u_out, u32 &h_out) {
  const u32 R = hi - lo;
  u32 qa, qb;
  if (V == 0) { qa = mulfrac_a(R, ghi0, ghi1); qb = mulfrac_a(R, glo0, glo1); }
  else { qa = mulfrac_b(R, ghi0, ghi1); qb = mulfrac_b(R, glo0, glo1); }
  const u32 nhi = ((ghi0 | ghi1) == 0) ? hi : lo + qa - 1;
  lo = lo + qb;
  hi = nhi;
  if (V <= 1) {
    const u32 x = lo ^ hi;
    const u32 k = x ? (u32)__clz(x) : 32u;
    h_out = hi;
    if (k == 32) { lo = 0; hi = 0xFFFFFFFFu; } else { lo <<= k; hi = (hi << k) | ((1u << k) - 1); }
    const u32 y = (lo & ~hi) << 1;
    const u32 u = (u32)__clz(~y);
    lo = ((lo << u) & 0x7FFFFFFFu) | (u ? 0u : (lo & 0x80000000u));
    hi = u ? ((hi << u) | ((1u << u) - 1) | 0x80000000u) : hi;
    k_out = k; u_out = u;
  } else {  // merged shift: t = k + u in one go
    const u32 x = lo ^ hi;
    const u32 k = x ? (u32)__clz(x) : 32u;
    h_out = hi;
    const u32 nh = ~hi;
    const u32 z = (u32)(((u64)(lo & nh)) << (k + 1));
    const u32 u = (u32)__clz(~z);
    const u32 t = k + u;
    lo = (u32)((u64)lo << t) & 0x7FFFFFFFu;
    hi = 0x80000000u | ~(u32)((u64)nh << t);
    k_out = k; u_out = u;
  }
}

template <int V> __global__ __launch_bounds__(64) void chain_scalar(u32 *out, int iters, uint4 seed) {
  const int lane = threadIdx.x;
  // operands: valid-looking reciprocal fractions, different per lane
  uint4 ops;
  ops.x = seed.x * (lane + 1) * 2654435761u; ops.y = (seed.y + lane * 7919u) & 0x3FFFFFFFu;
  ops.z = seed.z * (lane + 3) * 40503u; ops.w = ops.y + 0x20000000u + lane * 65537u;
  u32 lo = 0, hi = 0xFFFFFFFFu, acc = 0;
  for (int it = 0; it < iters; it++) {
    u32 rH = 0, rK = 0;
    for (u32 j = 0; j < 64; j++) {
      const u32 a0 = __builtin_amdgcn_readlane(ops.x, j), a1 = __builtin_amdgcn_readlane(ops.y, j);
      const u32 b0 = __builtin_amdgcn_readlane(ops.z, j), b1 = __builtin_amdgcn_readlane(ops.w, j);
      u32 k, u, h;
      step<V>(lo, hi, a0, a1, b0, b1, k, u, h);
      const bool mine = (u32)lane == j;
      rH = mine ? h : rH;
      rK = mine ? (k | (u << 8)) : rK;
    }
    acc += rH ^ rK;
  }
  out[blockIdx.x * 64 + lane] = acc + lo + hi;
}

// lanes-as-blocks: every lane runs its own chain (all VALU)
template <int V> __global__ __launch_bounds__(64) void chain_vector(u32 *out, int iters, uint4 seed) {
  const int lane = threadIdx.x;
  u32 lo = lane, hi = 0xFFFFFFFFu - lane, acc = 0;
  u32 a0 = seed.x * (lane + 1) * 2654435761u, a1 = (seed.y + lane * 7919u) & 0x3FFFFFFFu;
  u32 b0 = seed.z * (lane + 3) * 40503u, b1 = a1 + 0x20000000u + lane * 65537u;
  for (int it = 0; it < iters * 64; it++) {
    u32 k, u, h;
    step<V>(lo, hi, a0, a1, b0, b1, k, u, h);
    acc += h ^ k ^ u;
    a0 += 0x9E3779B9u; b0 += 0x7F4A7C15u;
  }
  out[blockIdx.x * 64 + lane] = acc + lo + hi;
}

template <typename K> void run(const char *name, K k, int nb, u32 *d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint4 seed = make_uint4(12345, 0x1234567, 777, 0);
  hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, d, 64, seed);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, d, iters, seed);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-22s blocks=%5d  %8.3f ms  %7.2f ns/symbol-step\n", name, nb, ms, ms * 1e6 / ((double)iters * 64));
}
int main() {
  setvbuf(stdout, 0, _IONBF, 0);
  u32 *d; hipMalloc(&d, 1 << 24);
  const int iters = 1 << 14;  // 1M symbols per block
  for (int nb : {1, 480, 1024, 2048, 4096}) {
    run("scalar V0 (current)", chain_scalar<0>, nb, d, iters);
    run("scalar V1 (mulfrac32)", chain_scalar<1>, nb, d, iters);
    run("scalar V2 (+merged)", chain_scalar<2>, nb, d, iters);
    run("vector V0", chain_vector<0>, nb, d, iters);
    run("vector V1", chain_vector<1>, nb, d, iters);
    run("vector V2", chain_vector<2>, nb, d, iters);
  }
  return 0;
}
