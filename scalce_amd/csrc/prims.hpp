// prims.hpp -- device primitives shared by the stage kernels: wave/block scans, a generic
// reduce-then-scan exclusive prefix sum, and one stable 8-bit LSD radix pass built on
// wavefront ballots (64-wide) with per-wave digit counters in LDS.
// gfx950 only: wave size is hard-coded to 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scalce {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;
// Pointers that a kernel loads from memory (descriptor tables) are generic to the compiler: it then emits flat_*
// instructions, which count on BOTH vmcnt and lgkmcnt, so that every LDS wait also waits for the global loads in
// flight.  Such pointers are cast to the global address space by hand.
#define SCALCE_GLOBAL __attribute__((address_space(1)))
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
// Workgroup barrier for waves that only exchange data through LDS: __syncthreads() is a fence as well and drains the
// global loads and stores in flight (s_waitcnt vmcnt(0)) in front of every barrier -- no prefetch survives it.
// (One asm statement with a memory clobber: to the compiler the s_barrier builtin touches no memory, so on its own it
// lets LDS reads of the next super-round be hoisted above the barrier -- a race that shows as a block coded wrongly
// once in a few runs.)
__device__ __forceinline__ void barrier_lds_only() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T t = __shfl_up(v, d, 64);
    if (lane_id() >= d) v += t;
  }
  return v;
}
// 32-bit sums stay in the DPP network: four shifted adds inside each 16-lane row, then the row totals travel with
// row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3).  Six v_add_u32_dpp instead of six
// ds_bpermute round trips through the LDS crossbar (each ~100 cycles of dependent latency).
template <>
__device__ __forceinline__ u32 wave_inclusive_sum<u32>(u32 v) {
  v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xF, 0xF, true);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xF, 0xF, true);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xF, 0xF, true);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xF, 0xF, true);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xA, 0xF, true);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xC, 0xF, true);  // row_bcast:31 -> rows 2, 3
  return v;
}

// Exclusive prefix over a block of NW waves.  smem must hold NW entries.  Ends with a barrier,
// so smem may be reused right after.
template <typename T, int NW>
__device__ __forceinline__ T block_exclusive_sum(T v, T *total, T *smem) {
  const int w = wave_id();
  T inc = wave_inclusive_sum(v);
  if (lane_id() == 63) smem[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    T s = smem[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// ---------------------------------------------------------------------------------------------
// exclusive scan: out(i) = sum_{j<i} f(j).  Three launches: per-tile sums, spine, downsweep.
// A workgroup's tile is SCAN_TILE consecutive elements.  Memory is touched in stripes -- instruction i of a thread reads
// element i * SCAN_THREADS + thread of the tile, so a wavefront's access is one contiguous run -- while the prefix wants
// every thread to own SCAN_ITEMS consecutive elements: the downsweep turns the tile through LDS (one spare slot per 16
// keeps both views free of bank conflicts) on the way in and on the way out.  (With each thread reading its own 16
// consecutive elements from memory, a wavefront's load touched 64 different cache lines per instruction and a scan of
// 61 M eight-byte keys ran at 0.45 TB/s.)
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
__device__ __forceinline__ int scan_slot(int i) { return i + (i >> 4); }

template <typename T, typename F>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_k(F f, u64 n, T *tile_sums) {
  __shared__ T sm[4];
  const u64 base = (u64)blockIdx.x * SCAN_TILE + threadIdx.x;
  T s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++)
    if (base + (u64)i * SCAN_THREADS < n) s += f(base + (u64)i * SCAN_THREADS);
  T tot;
  block_exclusive_sum<T, 4>(s, &tot, sm);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

template <typename T>
__global__ __launch_bounds__(1024) void scan_spine_k(T *sums, u32 nb, T *total_out) {
  __shared__ T sm[16];
  T carry = 0;
  for (u32 base = 0; base < nb; base += 1024) {
    const u32 i = base + threadIdx.x;
    T v = i < nb ? sums[i] : T(0);
    T tot;
    T ex = block_exclusive_sum<T, 16>(v, &tot, sm);
    if (i < nb) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename T, typename F, typename O>
__global__ __launch_bounds__(SCAN_THREADS) void scan_down_k(F f, u64 n, const T *tile_offs, O out) {
  __shared__ T tile[SCAN_TILE + SCAN_TILE / 16];
  __shared__ T sm[4];
  const u64 base = (u64)blockIdx.x * SCAN_TILE;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const int e = i * SCAN_THREADS + t;
    tile[scan_slot(e)] = (base + e < n) ? f(base + e) : T(0);
  }
  __syncthreads();
  T v[SCAN_ITEMS];
  T s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    v[i] = tile[scan_slot(t * SCAN_ITEMS + i)];
    s += v[i];
  }
  T tot;
  T run = tile_offs[blockIdx.x] + block_exclusive_sum<T, 4>(s, &tot, sm);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {  // (a thread's own slots: nobody else reads them before the barrier)
    tile[scan_slot(t * SCAN_ITEMS + i)] = run;
    run += v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    const int e = i * SCAN_THREADS + t;
    if (base + e < n) out(base + e, tile[scan_slot(e)]);
  }
}

// Host driver.  tile_ws must hold ceil(n / SCAN_TILE) elements of T.  total_out (device) may be
// null.  n == 0 writes *total_out = 0 through the spine kernel.
template <typename T, typename F, typename O>
inline void exclusive_scan(F f, u64 n, O out, T *tile_ws, T *total_out, hipStream_t st) {
  const u32 nb = (u32)((n + SCAN_TILE - 1) / SCAN_TILE);
  if (nb) hipLaunchKernelGGL((scan_reduce_k<T, F>), dim3(nb), dim3(SCAN_THREADS), 0, st, f, n, tile_ws);
  hipLaunchKernelGGL((scan_spine_k<T>), dim3(1), dim3(1024), 0, st, tile_ws, nb, total_out);
  if (nb) hipLaunchKernelGGL((scan_down_k<T, F, O>), dim3(nb), dim3(SCAN_THREADS), 0, st, f, n, tile_ws, out);
}

template <typename S, typename T>
struct LoadAs {
  const S *p;
  __device__ T operator()(u64 i) const { return (T)p[i]; }
};
template <typename T>
struct StoreTo {
  T *p;
  __device__ void operator()(u64 i, T v) const { p[i] = v; }
};

// ---------------------------------------------------------------------------------------------
// one stable LSD radix pass over an array of u32 payloads; the 8-bit digit of a payload is
// given by a functor (it may gather from anywhere).  hist layout: [256][ntiles] so that a flat
// exclusive scan yields the global base of (digit, tile).
// ---------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;

template <typename D>
__global__ __launch_bounds__(RS_THREADS) void radix_hist_k(const u32 *in, u32 n, D digit, u32 *hist, u32 ntiles) {
  __shared__ u32 h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const u32 base = blockIdx.x * RS_TILE;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; i++) {
    const u32 idx = base + i * RS_THREADS + threadIdx.x;
    if (idx < n) atomicAdd(&h[digit(in ? in[idx] : idx)], 1u);
  }
  __syncthreads();
  hist[(u64)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

template <typename D>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_k(const u32 *in, u32 *out, u32 n, D digit,
                                                             const u32 *offs, u32 ntiles) {
  __shared__ u32 wh[4][256];
  for (int i = threadIdx.x; i < 4 * 256; i += RS_THREADS) (&wh[0][0])[i] = 0;
  __syncthreads();
  const int w = wave_id(), lane = lane_id();
  const u64 lt = (1ull << lane) - 1;
  const u32 base = blockIdx.x * RS_TILE + w * (64 * RS_ITEMS);
  u32 val[RS_ITEMS], pos[RS_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const u32 idx = base + r * 64 + lane;
    const bool valid = idx < n;
    const u32 v = valid ? (in ? in[idx] : idx) : 0u;
    const u32 d = valid ? (u32)digit(v) : 0u;
    u64 peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1;
      const u64 bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const u32 rank = __popcll(peers & lt);
    u32 pre = 0;
    if (valid && rank == 0) {  // lowest lane of each digit group bumps this wave's counter
      pre = wh[w][d];
      wh[w][d] = pre + (u32)__popcll(peers);
    }
    const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
    pre = __shfl(pre, leader, 64);
    val[r] = v;
    pos[r] = valid ? ((d << 16) | (pre + rank)) : 0xFFFFFFFFu;  // tile <= 2048 items: rank fits 16 bits
  }
  __syncthreads();
  {
    const u32 d = threadIdx.x;
    const u32 c0 = wh[0][d], c1 = wh[1][d], c2 = wh[2][d];
    const u32 g = offs[(u64)d * ntiles + blockIdx.x];
    wh[0][d] = g;
    wh[1][d] = g + c0;
    wh[2][d] = g + c0 + c1;
    wh[3][d] = g + c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++)
    if (pos[r] != 0xFFFFFFFFu) out[wh[w][pos[r] >> 16] + (pos[r] & 0xFFFFu)] = val[r];
}

// in == nullptr means the identity permutation 0..n-1.  hist_ws: 256*ntiles u32; tile_ws: scan
// workspace for 256*ntiles elements.
template <typename D>
inline void radix_pass(const u32 *in, u32 *out, u32 n, D digit, u32 *hist_ws, u32 *tile_ws, hipStream_t st) {
  if (!n) return;
  const u32 ntiles = (n + RS_TILE - 1) / RS_TILE;
  hipLaunchKernelGGL((radix_hist_k<D>), dim3(ntiles), dim3(RS_THREADS), 0, st, in, n, digit, hist_ws, ntiles);
  exclusive_scan<u32>(LoadAs<u32, u32>{hist_ws}, (u64)256 * ntiles, StoreTo<u32>{hist_ws}, tile_ws, (u32 *)nullptr, st);
  hipLaunchKernelGGL((radix_scatter_k<D>), dim3(ntiles), dim3(RS_THREADS), 0, st, in, out, n, digit, hist_ws, ntiles);
}

// ---------------------------------------------------------------------------------------------
// the same pass over (64-bit key, u32 payload) pairs: the digit is byte `shift / 8` of the key that
// travels with the payload, so every read is sequential (radix_pass's functor gathers through the
// payload: 8 GB of sector fetches per pass for 50 M two-byte digits)
// ---------------------------------------------------------------------------------------------
// A workgroup counts RH_TILES consecutive tiles: the table is digit-major (hist[digit][tile], what the scan wants), so a
// workgroup of one tile wrote 256 single words a whole row apart -- 64 bytes of HBM write per word.  Sixteen tiles make
// every digit's words one 64-byte run.
constexpr int RH_TILES = 16;
template <typename K>
__global__ __launch_bounds__(RS_THREADS) void radix_hist_key_k(const K *keys, u32 n, u32 shift, u32 *hist, u32 ntiles) {
  __shared__ u32 h[RH_TILES][256];
  for (int i = threadIdx.x; i < RH_TILES * 256; i += RS_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const u32 tile0 = blockIdx.x * RH_TILES;
  for (int j = 0; j < RH_TILES; j++) {
    const u32 base = (tile0 + j) * RS_TILE;
    if (base >= n) break;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
      const u32 idx = base + i * RS_THREADS + threadIdx.x;
      if (idx < n) atomicAdd(&h[j][(u32)(keys[idx] >> shift) & 0xFFu], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RH_TILES * 256; i += RS_THREADS) {  // sixteen lanes = the sixteen tiles of one digit
    const u32 d = (u32)i / RH_TILES, j = (u32)i % RH_TILES;
    if (tile0 + j < ntiles) hist[(u64)d * ntiles + tile0 + j] = h[j][d];
  }
}
// vals_in == nullptr means the identity permutation
// The same scatter through LDS: the tile's pairs are first put in digit order in LDS, then leave in that order, so that a
// wavefront's store instruction covers a few runs of consecutive addresses (8 pairs per digit and tile on average) instead
// of 64 single pairs -- WRITE_SIZE of the direct version was 1.9 x the bytes it stores.
// 512 threads on a tile of 2048 pairs: the staging buffers (37 KB) allow four workgroups per CU, and four items per thread
// with eight waves each keep 32 waves resident where 256 threads kept 16.
constexpr int RSK_THREADS = 512;
constexpr int RSK_ITEMS = RS_TILE / RSK_THREADS;
constexpr int RSK_WAVES = RSK_THREADS / 64;
template <typename K>
__global__ __launch_bounds__(RSK_THREADS) void radix_scatter_kv_staged_k(const K *keys_in, const u32 *vals_in, K *keys_out, u32 *vals_out,
                                                                        u32 n, u32 shift, const u32 *offs, u32 ntiles) {
  __shared__ K keys_s[RS_TILE];
  __shared__ u32 vals_s[RS_TILE], dest_s[RS_TILE];
  __shared__ u32 wh[RSK_WAVES][256];
  __shared__ u32 gbase[256];
  __shared__ u32 sm[RSK_WAVES];
  for (int i = threadIdx.x; i < RSK_WAVES * 256; i += RSK_THREADS) (&wh[0][0])[i] = 0;
  __syncthreads();
  const int w = wave_id(), lane = lane_id();
  const u64 lt = (1ull << lane) - 1;
  const u32 tile0 = blockIdx.x * RS_TILE, base = tile0 + w * (64 * RSK_ITEMS);
  K key[RSK_ITEMS];
  u32 val[RSK_ITEMS], pos[RSK_ITEMS];
#pragma unroll
  for (int r = 0; r < RSK_ITEMS; r++) {
    const u32 idx = base + r * 64 + lane;
    const bool valid = idx < n;
    key[r] = valid ? keys_in[idx] : K(0);
    val[r] = valid ? (vals_in ? vals_in[idx] : idx) : 0u;
    const u32 d = (u32)(key[r] >> shift) & 0xFFu;
    u64 peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1;
      const u64 bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const u32 rank = __popcll(peers & lt);
    u32 pre = 0;
    if (valid && rank == 0) {
      pre = wh[w][d];
      wh[w][d] = pre + (u32)__popcll(peers);
    }
    const int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
    pre = __shfl(pre, leader, 64);
    pos[r] = valid ? ((d << 16) | (pre + rank)) : 0xFFFFFFFFu;
  }
  __syncthreads();
  {
    // digit d of the tile starts behind all smaller digits; inside it the waves follow each other
    u32 tot = 0;
    const u32 d = threadIdx.x;
    u32 c[RSK_WAVES];
    if (d < 256) {
#pragma unroll
      for (int i = 0; i < RSK_WAVES; i++) { c[i] = wh[i][d]; tot += c[i]; }
    }
    u32 all;
    const u32 ex = block_exclusive_sum<u32, RSK_WAVES>(d < 256 ? tot : 0u, &all, sm);
    if (d < 256) {
      u32 run = ex;
#pragma unroll
      for (int i = 0; i < RSK_WAVES; i++) { wh[i][d] = run; run += c[i]; }
      gbase[d] = offs[(u64)d * ntiles + blockIdx.x] - ex;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RSK_ITEMS; r++)
    if (pos[r] != 0xFFFFFFFFu) {
      const u32 d = pos[r] >> 16, l = wh[w][d] + (pos[r] & 0xFFFFu);
      keys_s[l] = key[r];
      vals_s[l] = val[r];
      dest_s[l] = gbase[d] + l;
    }
  __syncthreads();
  const u32 cnt = n - tile0 < (u32)RS_TILE ? n - tile0 : (u32)RS_TILE;
#pragma unroll
  for (int r = 0; r < RSK_ITEMS; r++) {
    const u32 j = r * RSK_THREADS + threadIdx.x;
    if (j < cnt) {
      const u32 at = dest_s[j];
      keys_out[at] = keys_s[j];
      vals_out[at] = vals_s[j];
    }
  }
}

template <typename K>
inline void radix_pass_kv(const K *keys_in, const u32 *vals_in, K *keys_out, u32 *vals_out, u32 n, u32 shift, u32 *hist_ws,
                          u32 *tile_ws, hipStream_t st) {
  if (!n) return;
  const u32 ntiles = (n + RS_TILE - 1) / RS_TILE;
  hipLaunchKernelGGL(radix_hist_key_k<K>, dim3((ntiles + RH_TILES - 1) / RH_TILES), dim3(RS_THREADS), 0, st, keys_in, n, shift, hist_ws, ntiles);
  exclusive_scan<u32>(LoadAs<u32, u32>{hist_ws}, (u64)256 * ntiles, StoreTo<u32>{hist_ws}, tile_ws, (u32 *)nullptr, st);
  hipLaunchKernelGGL(radix_scatter_kv_staged_k<K>, dim3(ntiles), dim3(RSK_THREADS), 0, st, keys_in, vals_in, keys_out, vals_out, n, shift,
                     hist_ws, ntiles);
}

inline u64 scan_ws_elems(u64 n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }
inline u64 radix_hist_elems(u64 n) { return 256 * ((n + RS_TILE - 1) / RS_TILE) + 256; }

}  // namespace scalce
