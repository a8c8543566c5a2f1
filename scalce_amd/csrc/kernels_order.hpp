// kernels_order.hpp -- bucket/reorder and stream emission.
//   order: the permutation aho_output + bin_prepare produce (/root/reference/reads.cpp:466-499,
//          547-634): buckets by BFS id, root last; inside a bucket, spill chunks in order
//          (merhamet_merge, compress.cpp:104-159); inside a chunk a stable sort on the suffix that
//          follows the core, right-padded with A (_POS, reads.cpp:557-558).
//          Done as LSD radix passes (prims.hpp) over u32 read indices: key digits from the least
//          significant, then the chunk id, then the bucket id.
//   emit:  rotated 2-bit records with their end marker and the per-bucket headers of .scalcer
//          (output_read reads.cpp:432-461, bin_dump :114-131, compress.cpp:364-379), the name
//          stream, and the quality stream in output order.
#pragma once
#include "kernels_token.hpp"

namespace scalce {

// digit d of the sort key of read v: stored bases 4d..4d+3 of the rotated read, i.e. original
// bases end+4d .. end+4d+3, each replaced by 0 (A) once it runs past the read (reads.cpp:557-558)
struct KeyDigit {
  const u8 *packed;
  const u16 *end;
  int L, stride, d;
  __device__ u32 operator()(u32 v) const {
    const int s = (int)end[v] + 4 * d;  // first original base of this digit
    if (s >= L) return 0u;
    const u8 *row = packed + (u64)v * stride;
    const int byte = s >> 2, sh = (s & 3) * 2;
    u32 w = ((u32)row[byte] << 8) | (u32)row[byte + 1];  // rows are zero padded past SZ_READ(L)
    u32 dig = (w >> (8 - sh)) & 0xFFu;
    const int valid = L - s;  // bases of this digit that exist
    if (valid < 4) dig &= (0xFFu << (2 * (4 - valid))) & 0xFFu;
    return dig;
  }
};

// ---- two-phase order -----------------------------------------------------------------------------
// Phase 1 sorts by (bucket, chunk, first PREFIX_DIGITS key digits); records that still tie with a neighbour
// form "runs" and only those go through the remaining key digits (phase 2).  On reads without exact
// duplicates almost nothing is left for phase 2, so the random digit gathers drop from ceil(L/4) passes to
// PREFIX_DIGITS.
constexpr int PREFIX_DIGITS = 4;            // (three digits -- five passes instead of six -- were tried in round 4: runs of more than 32 records that agree
                                            //  on 12 key bases are common enough to send every 50 M-read shard through the 22 radix passes of phase 2: +12 ms)
constexpr int PREFIX_BITS = 8 * PREFIX_DIGITS;

struct RunArgs {
  u32 n;
  const u32 *perm;      // order after phase 1
  const u32 *bucket;
  const u32 *chunk;     // or null
  const u8 *packed;
  const u16 *end;
  int L, stride, ndig1; // digits sorted in phase 1
};
__device__ __forceinline__ bool same_phase1_key(const RunArgs &a, u32 x, u32 y) {
  if (a.bucket[x] != a.bucket[y]) return false;
  if (a.chunk && a.chunk[x] != a.chunk[y]) return false;
  for (int d = 0; d < a.ndig1; d++) {
    KeyDigit kd{a.packed, a.end, a.L, a.stride, d};
    if (kd(x) != kd(y)) return false;
  }
  return true;
}
// head[i] = 1 when position i starts a new phase-1 key
__global__ __launch_bounds__(256) void run_heads_k(RunArgs a, u8 *head) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  head[i] = (i == 0) ? 1 : (same_phase1_key(a, a.perm[i - 1], a.perm[i]) ? 0 : 1);
}
// Phase 1 on (key, read) pairs: the key is what phase 1 sorts by, most significant first --
// bucket | chunk | first PREFIX_DIGITS digits -- built once from the rows in input order (sequential reads)
__global__ __launch_bounds__(256) void order_keys_k(u32 n, const u32 *bucket, const u32 *chunk /* or null */, u32 chunk_bits,
                                                   const u8 *packed, const u16 *end, int L, int stride, int ndig1, u32 end_bits,
                                                   u64 *keys) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 prefix = 0;
  for (int d = 0; d < PREFIX_DIGITS; d++) {
    KeyDigit kd{packed, end, L, stride, d};
    prefix = (prefix << 8) | (d < ndig1 ? kd(i) : 0u);
  }
  u64 hi = bucket[i];
  if (chunk) hi = (hi << chunk_bits) | chunk[i];
  // where 16 bits are left below the sorted part, the record's `end` rides along: the emit stage then finds bucket and
  // end of the k-th record in the k-th key instead of gathering them through the permutation
  keys[i] = (((hi << PREFIX_BITS) | prefix) << end_bits) | (end_bits ? (u64)end[i] : 0ull);
}
__global__ __launch_bounds__(256) void run_heads_keys_k(u32 n, const u64 *sorted_keys, u32 end_bits, u8 *head) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || (sorted_keys[i] >> end_bits) != (sorted_keys[i - 1] >> end_bits)) ? 1 : 0;
}
struct RunMember {  // position i belongs to a run of length > 1
  const u8 *head;
  u32 n;
  __device__ u32 operator()(u64 i) const { return (!head[i] || (i + 1 < n && !head[i + 1])) ? 1u : 0u; }
};
// compaction of the run members: items (read indices), their positions, and the run id of every member
__global__ __launch_bounds__(256) void run_compact_k(u32 n, const u8 *head, const u32 *member_rank, const u32 *head_count,
                                                    const u32 *perm, u32 *items, u32 *pos_list, u32 *runid_of_read) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool member = !head[i] || (i + 1 < n && !head[i + 1]);
  if (!member) return;
  const u32 m = member_rank[i];
  const u32 r = perm[i];
  items[m] = r;
  pos_list[m] = i;
  runid_of_read[r] = head_count[i] + head[i];  // heads at positions <= i: constant inside a run, increasing across runs
}
__global__ __launch_bounds__(256) void run_scatter_k(u32 m, const u32 *sorted, const u32 *pos_list, u32 *perm, u64 *keys /* or null */,
                                                    u32 end_bits, const u16 *end) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const u32 p = pos_list[i], r = sorted[i];
  perm[p] = r;
  // the members of a run share the sorted part of the key; the `end` riding below it follows the record
  if (keys && end_bits) keys[p] = ((keys[p] >> end_bits) << end_bits) | (u64)end[r];
}

// Phase 2 without passes.  On reads without many exact duplicates the runs are pairs and triples -- and long runs of
// records whose key ENDS inside the prefix (core near the end of the read: nothing is left to sort them by, they are in
// their final order already).  25 radix passes (21 remaining key digits + the run id) over 1.7 M members are 125 small
// launches, 1.9 ms of launch latency.  Here the thread of a run's first member sorts a run of up to RUN_SMALL_MAX members
// where it stands -- insertion sort on the remaining digits, most significant first, strictly-less so that equal keys keep
// their phase-1 (= input) order.  Longer runs are left alone; a member of one that HAS bases behind the prefix sets
// `any_large`, and the host then takes the radix passes for everything (heavy duplicates).
constexpr u32 RUN_SMALL_MAX = 32;
__global__ __launch_bounds__(256) void run_small_sort_k(u32 M, const u32 *pos_list, const u8 *head, u32 n, u32 *perm, u64 *keys /* or null */,
                                                       u32 end_bits, const u8 *packed, const u16 *end, int L, int stride, int ndig1,
                                                       int ndig, u32 *any_large) {
  const u32 m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const u32 p = pos_list[m];
  if (!head[p]) {  // not the first member of its run: does it make a long run one that needs sorting?
    if ((int)end[perm[p]] + 4 * ndig1 >= L) return;       // no bases behind the prefix
    u32 back = 1;
    while (back <= RUN_SMALL_MAX && !head[p - back]) back++;   // (p - back is the run's first position when the loop ends on a head)
    u32 fwd = 1;
    while (fwd <= RUN_SMALL_MAX && p + fwd < n && !head[p + fwd]) fwd++;
    if (back + fwd > RUN_SMALL_MAX) atomicOr(any_large, 1u);
    return;
  }
  u32 len = 1;
  while (p + len < n && !head[p + len] && len <= RUN_SMALL_MAX) len++;
  if (len > RUN_SMALL_MAX) {
    if ((int)end[perm[p]] + 4 * ndig1 < L) atomicOr(any_large, 1u);
    return;
  }
  auto less = [&](u32 x, u32 y) -> bool {
    for (int d = ndig1; d < ndig; d++) {
      const u32 dx = KeyDigit{packed, end, L, stride, d}(x), dy = KeyDigit{packed, end, L, stride, d}(y);
      if (dx != dy) return dx < dy;
    }
    return false;
  };
  for (u32 a = 1; a < len; a++) {
    const u32 x = perm[p + a];
    u32 j = a;
    while (j > 0 && less(x, perm[p + j - 1])) { perm[p + j] = perm[p + j - 1]; j--; }
    perm[p + j] = x;
  }
  if (keys && end_bits)  // the members of a run share the sorted part of the key; the `end` riding below it follows the record
    for (u32 a = 0; a < len; a++) keys[p + a] = ((keys[p + a] >> end_bits) << end_bits) | (u64)end[perm[p + a]];
}

// spill chunks (compress.cpp:702-715): running size of the records since the last dump; when it
// reaches -B the current read closes the chunk.  rec_size is scanned inclusively into S; the
// boundaries are found by one thread with binary searches (there are few chunks).
struct RecSize {
  const u32 *bucket;
  const u32 *bucket_level;
  const u8 *namelen;
  int L0, L1, paired, use_names, has_qual;
  __device__ u64 operator()(u64 r) const {
    const int lv = (int)bucket_level[bucket[r]];
    u64 sz = (use_names ? 1u + namelen[r] : 1u) + (u64)((L0 - lv + 3) >> 2) + (has_qual ? L0 : 0);
    if (paired) sz += (u64)((L1 + 3) >> 2) + (has_qual ? L1 : 0);
    return sz + 40;  // + sizeof(bin_node), compress.cpp:702
  }
};
__global__ void chunk_bounds_k(const u64 *S /*exclusive prefix, S[n] = total*/, u64 n, u64 limit, u32 max_chunks,
                               u64 *chunk_start /*[max_chunks+1]*/, u32 *nchunks) {
  if (threadIdx.x || blockIdx.x) return;
  u32 c = 0;
  u64 start = 0;
  chunk_start[0] = 0;
  while (start < n && c + 1 < max_chunks) {
    // smallest r >= start with S[r+1] - S[start] >= limit; the chunk is [start, r]
    const u64 need = S[start] + limit;
    u64 lo = start, hi = n;  // search r in [start, n)
    while (lo < hi) {
      const u64 mid = (lo + hi) >> 1;
      if (S[mid + 1] >= need) hi = mid; else lo = mid + 1;
    }
    if (lo >= n) break;  // the tail never reaches the limit: last chunk runs to n
    start = lo + 1;
    chunk_start[++c] = start;
  }
  if (chunk_start[c] < n || c == 0) c++;  // trailing partial chunk (compress.cpp:799-801)
  chunk_start[c] = n;
  *nchunks = c;
}
// The same rule for a rank of a sharded run: `carry_in` bytes are already in the chunk that is open when this rank's rows
// begin (the rows of the ranks before it), cuts[] = rows in front of which a new chunk starts (1 .. n), carry_out = bytes
// in the chunk still open behind the last row.
__global__ void chunk_cuts_k(const u64 *S /*exclusive prefix, S[n] = total*/, u64 n, u64 limit, u64 carry_in, u32 max_cuts,
                             u64 *cuts, u32 *ncuts, u64 *carry_out) {
  if (threadIdx.x || blockIdx.x) return;
  u32 c = 0;
  u64 start = 0, carry = carry_in;
  while (limit && start < n && c < max_cuts) {
    const u64 need = S[start] + (limit > carry ? limit - carry : 0);
    u64 lo = start, hi = n;  // smallest r in [start, n) with S[r + 1] >= need
    while (lo < hi) {
      const u64 mid = (lo + hi) >> 1;
      if (S[mid + 1] >= need) hi = mid; else lo = mid + 1;
    }
    if (lo >= n) break;
    start = lo + 1;
    cuts[c++] = start;
    carry = 0;
  }
  *carry_out = carry + (S[n] - S[start]);
  *ncuts = c;
}
__global__ __launch_bounds__(256) void chunk_assign_k(u64 n, const u64 *chunk_start, const u32 *nchunks, u32 *chunk) {
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  u32 lo = 0, hi = *nchunks;  // largest c with chunk_start[c] <= r
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (chunk_start[mid] <= r) lo = mid; else hi = mid;
  }
  chunk[r] = lo;
}

// ---- emission ---------------------------------------------------------------------------------
// per bucket: first output position and byte offset of its header in the .scalcer payload
struct BucketBytes {
  const u64 *counts;
  const u32 *bucket_level;
  int L, sz_meta;
  __device__ u64 operator()(u64 b) const {
    const u64 c = counts[b];
    return c ? 12 + c * (u64)(((L - (int)bucket_level[b] + 3) >> 2) + sz_meta) : 0;
  }
};

struct EmitArgs {
  u64 nrec;
  const u32 *perm;
  const u32 *bucket;
  const u16 *end;
  const u8 *packed;
  int L, stride, sz_meta;
  const u32 *bucket_level;
  const int32_t *bucket_pattern;
  const u64 *bucket_first;  // exclusive scan of counts
  const u64 *bucket_off;    // exclusive scan of BucketBytes
  const u64 *counts;
  u8 *out;
  const u64 *keys;          // sorted phase-1 keys (position k <-> record perm[k]) or null
  u32 key_bucket_shift, key_bucket_mask, key_end_bits;
  // Fused rows (emit_reads_k<true>): `stride` is the row's size, the copy of the packed words (`pwords` of them) lies
  // cell_off bytes into the row (behind its q'); the same workgroup also takes the row's q' (frow, L bytes, L % 4 == 0) into
  // the reordered stream `qs` -- one random line per record for the bases and the qualities.  (cells_sorted: a name cell
  // cell_off + 4 * pwords bytes into the row, for layouts that carry one; the default layout does not.)
  int pwords;
  const u8 *frow;
  u32 cell_off, qunits;     // qunits = ceil(L / 16): 16-byte units of a row's q'
  u64 qmagic, rmagic, lmagic;  // ceil(2^32 / qunits), ceil(2^32 / (stride / 16)), ceil(2^32 / L)
  u8 *cells_sorted, *outlen, *qs;
};
// The records of a workgroup's 256 positions are contiguous in the output (bucket headers included), 27 bytes each at
// L = 100 and not aligned to anything: written by their own threads byte by byte they cost 4.6 bytes of HBM writes per
// byte (PMC WRITE_SIZE).  They are assembled in LDS instead and leave as one coalesced block.
constexpr int EMIT_STAGE_BYTES = 256 * 64;
template <bool FUSED>
__global__ __launch_bounds__(256) void emit_reads_k(EmitArgs a) {
  __shared__ __attribute__((aligned(16))) u8 stage[EMIT_STAGE_BYTES];
  __shared__ u64 wg_begin, wg_end;
  __shared__ u32 s_row[FUSED ? 256 : 1];
  const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = k < a.nrec;
  u64 my_begin = 0, my_end = 0, rec_at = 0;
  u32 r = 0, b = 0;
  int lv = 0, e = 0, recsz = 0;
  u64 first = 0;
  if (live) {
    r = a.perm[k];
    const u64 key = a.keys ? a.keys[k] : 0ull;
    b = a.keys ? ((u32)(key >> a.key_bucket_shift) & a.key_bucket_mask) : a.bucket[r];
    lv = (int)a.bucket_level[b];
    e = (a.keys && a.key_end_bits) ? (int)(key & 0xFFFFu) : (int)a.end[r];
    recsz = ((a.L - lv + 3) >> 2) + a.sz_meta;
    first = a.bucket_first[b];
    rec_at = a.bucket_off[b] + 12 + (k - first) * (u64)recsz;
    my_begin = (k == first) ? a.bucket_off[b] : rec_at;
    my_end = rec_at + (u64)recsz;
  }
  // Fused rows: the workgroup's 256 rows come into LDS first, whole -- every thread issues its share of the 16-byte pieces
  // back to back (consecutive threads take consecutive pieces of a row), so a row costs ONE trip to memory and all of them
  // are on their way at once.  (Read where they are used -- the packed words by the record's thread, the q' pieces behind a
  // barrier -- the same bytes took 8.5 ms per 50 M reads: 5.1 for the first touch of a row, 2.2 for its other line, 0.9 for
  // the cell, tools/emit_ablate.sh.)  Everything below then reads the row from LDS.
  extern __shared__ uint4 s_rows[];  // [256][stride / 16]
  const u32 rs16 = FUSED ? (u32)a.stride >> 4 : 0u;
  if (FUSED) {
    s_row[threadIdx.x] = r;
    __syncthreads();
    const u64 k0 = (u64)blockIdx.x * blockDim.x;
    const u32 nlive = (u32)(a.nrec - k0 < 256 ? a.nrec - k0 : 256);
    const u32 units = nlive * rs16;
    // straight into LDS (global_load_lds_dwordx4: lane l of a wave's instruction lands at the wave's base + 16 l, which is
    // unit u's place), no register in between.  (Through a register array -- loads first, LDS stores behind them -- the
    // compiler put the array in SCRATCH: 272 bytes per thread, 6.4 GB written to HBM and read back per 50 M-read shard;
    // found by WRITE_SIZE, tools/emit_write_probe.sh.)
    const u32 wave_base = (threadIdx.x & ~63u);
    for (u32 i = 0; i * 256u < units; i++) {
      const u32 u = i * 256u + threadIdx.x;
      if (u < units) {
        const u32 rec = (u32)(((u64)u * a.rmagic) >> 32), j = u - rec * rs16;
        const u8 *src = a.frow + (u64)s_row[rec] * a.stride + 16 * j;
        __builtin_amdgcn_global_load_lds((const SCALCE_GLOBAL void *)src, (__attribute__((address_space(3))) void *)(s_rows + i * 256u + wave_base), 16, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the rows are in LDS before the barrier below lets anyone read them)
  }
  if (threadIdx.x == 0) wg_begin = my_begin;
  if (live && (threadIdx.x == blockDim.x - 1 || k + 1 == a.nrec)) wg_end = my_end;
  __syncthreads();
  const u32 *lrow = FUSED ? reinterpret_cast<const u32 *>(s_rows + (u32)threadIdx.x * rs16) : nullptr;  // this thread's record's row
  if (FUSED && live && a.cells_sorted) {  // the name cell -> position k
    const u32 *c = lrow + (a.cell_off >> 2) + a.pwords;
    *reinterpret_cast<uint4 *>(a.cells_sorted + 16 * k) = make_uint4(c[0], c[1], c[2], c[3]);
    a.outlen[k] = (u8)(c[0] & 0xFFu);
  }
  const bool staged = wg_end - wg_begin <= (u64)EMIT_STAGE_BYTES;  // always, short of reads of more than ~200 bases
  u8 *dst = staged ? stage + (rec_at - wg_begin) : a.out + rec_at;
  if (live) {
    if (k == first) {  // [int32 core][int64 count], compress.cpp:365-378
      u8 *h = dst - 12;
      const u32 core = (u32)a.bucket_pattern[b];
      const u64 cnt = a.counts[b];
      for (int i = 0; i < 4; i++) h[i] = (u8)(core >> (8 * i));
      for (int i = 0; i < 8; i++) h[4 + i] = (u8)(cnt >> (8 * i));
    }
    // output_read(line, dest, n = e - lv, l = lv): bases [e, L) then [0, e - lv).  Done on 32-bit big-endian
    // words of the packed row: output word j = up to two bit-field fetches (funnel shifts) instead of 16
    // single-base extractions.
    const u32 *roww = FUSED ? lrow + (a.cell_off >> 2) : reinterpret_cast<const u32 *>(a.packed + (u64)r * a.stride);
    const int nwords = FUSED ? a.pwords : a.stride >> 2;
    auto S = [&](int i) -> u32 { return i < nwords ? __builtin_bswap32(roww[i]) : 0u; };
    auto bits32 = [&](int pos) -> u32 {  // 32 source bits starting at bit `pos` (MSB first)
      const int w = pos >> 5, sh = pos & 31;
      const u32 w0 = S(w);
      if (!sh) return w0;
      return (w0 << sh) | (S(w + 1) >> (32 - sh));
    };
    const int n = e ? e - lv : 0;
    const int A = 2 * (a.L - e);        // bits of the part after the core
    const int T = A + 2 * n;            // bits of the record (2 * (L - lv) with a core, 2 * L without)
    const int nbytes = (T + 7) >> 3;
    int j = 0;
    for (int p0 = 0; p0 < T; p0 += 32) {
      int n1 = A - p0;
      n1 = n1 < 0 ? 0 : (n1 > 32 ? 32 : n1);
      u32 word = 0;
      if (n1) word = bits32(2 * e + p0) & (0xFFFFFFFFu << (32 - n1));
      if (n1 < 32) word |= bits32(p0 + n1 - A) >> n1;
      const int left = T - p0;          // record bits from this word on
      if (left < 32) word &= 0xFFFFFFFFu << (32 - left);
#pragma unroll
      for (int t = 0; t < 4; t++)
        if (j < nbytes) dst[j++] = (u8)(word >> (24 - 8 * t));
    }
    dst[j] = (u8)e;  // end marker, reads.cpp:130
    if (a.sz_meta == 2) dst[j + 1] = (u8)(e >> 8);
  }
  if (FUSED && a.qs) {
    // q' of the workgroup's records: a contiguous, 16-byte aligned piece of the stream (256 x L bytes, L % 4 == 0).  A
    // thread produces aligned 16-byte chunks of it, every word from where it lies in the rows in LDS (copied record by
    // record in units of 16 bytes at 4-byte boundaries -- L = 100 -- the stores straddled sectors: WRITE_SIZE counted the
    // stream twice).
    const u64 k0 = (u64)blockIdx.x * blockDim.x;
    const u32 nlive = (u32)(a.nrec - k0 < 256 ? a.nrec - k0 : 256);
    const u32 L = (u32)a.L, total = nlive * L, rsb = (u32)a.stride;
    const u8 *lrows = reinterpret_cast<const u8 *>(s_rows);
    u8 *qdst = a.qs + k0 * (u64)L;
    for (u32 c = threadIdx.x * 16u; c < total; c += 256u * 16u) {
      u32 wv[4];
#pragma unroll
      for (u32 x = 0; x < 4; x++) {
        const u32 o = c + 4 * x;
        const u32 rec = (u32)(((u64)o * a.lmagic) >> 32), col = o - rec * L;   // o / L, o % L
        wv[x] = o < total ? *reinterpret_cast<const u32 *>(lrows + rec * rsb + col) : 0u;
      }
      if (c + 16 <= total) *reinterpret_cast<uint4 *>(qdst + c) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
      else for (u32 x = 0; c + 4 * x < total; x++) *reinterpret_cast<u32 *>(qdst + c + 4 * x) = wv[x];
    }
  }
  if (!staged) return;
  __syncthreads();
  // the block leaves: bytes up to the first 4-byte boundary of the output, whole words, the tail
  u8 *out = a.out + wg_begin;
  const u32 span = (u32)(wg_end - wg_begin);
  const u32 head = (u32)((4 - ((u64)out & 3)) & 3) < span ? (u32)((4 - ((u64)out & 3)) & 3) : span;
  if (threadIdx.x < head) out[threadIdx.x] = stage[threadIdx.x];
  const u32 nwords_out = (span - head) >> 2;
  for (u32 w = threadIdx.x; w < nwords_out; w += blockDim.x) {
    const u8 *sp = stage + head + 4 * w;
    const u32 v = (u32)sp[0] | ((u32)sp[1] << 8) | ((u32)sp[2] << 16) | ((u32)sp[3] << 24);
    *reinterpret_cast<u32 *>(out + head + 4 * w) = v;
  }
  const u32 done = head + 4 * nwords_out;
  if (threadIdx.x < span - done) out[done + threadIdx.x] = stage[done + threadIdx.x];
}

struct NameLenOut {  // bytes of the k-th emitted name record
  const u32 *perm;
  const u8 *namelen;
  __device__ u64 operator()(u64 k) const { return 1ull + namelen[perm[k]]; }
};
// the same, written down once in output order: the two scan passes then read it sequentially instead of gathering twice
__global__ __launch_bounds__(256) void name_outlen_k(u64 nrec, const u32 *perm, const u8 *namelen, u8 *outlen) {
  const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nrec) outlen[k] = namelen[perm[k]];
}
struct NameLenSeq {
  const u8 *outlen;
  __device__ u64 operator()(u64 k) const { return 1ull + outlen[k]; }
};
// bytes of the name records of every bucket (sharded runs: where a rank's names go in the run-wide stream)
__global__ __launch_bounds__(256) void bucket_name_bytes_k(u32 nb1, const u64 *bucket_first, const u64 *counts, const u64 *name_off,
                                                          const u64 *names_total, u64 nrec, u64 *out) {
  const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb1) return;
  const u64 f = bucket_first[b], e = f + counts[b];
  const u64 lo = f < nrec ? name_off[f] : *names_total, hi = e < nrec ? name_off[e] : *names_total;
  out[b] = counts[b] ? hi - lo : 0;
}

// The 16-byte cell written by the ingest stage holds the length and up to 15 characters: ONE gather per record; a longer
// name comes from the long-name store (input order, written when its piece was ingested).
__global__ __launch_bounds__(256) void emit_names_k(u64 nrec, const u32 *perm, const u8 *cells, const u64 *store_off,
                                                   const u8 *store, const u64 *name_off, u8 *out) {
  const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nrec) return;
  const u32 r = perm[k];
  u8 *dst = out + name_off[k];
  const uint4 c = *reinterpret_cast<const uint4 *>(cells + 16 * (u64)r);
  const u32 n = c.x & 0xFFu;
  if (n <= 15) {
    const u32 w[4] = {c.x, c.y, c.z, c.w};
    for (u32 i = 0; i <= n; i++) dst[i] = (u8)(w[i >> 2] >> (8 * (i & 3)));
    return;
  }
  const u8 *src = store + store_off[r];
  dst[0] = (u8)n;
  for (u32 i = 0; i < n; i++) dst[1 + i] = src[i];
}

// The same with the cells already in output order (name_cells_sorted_k): one gather through the permutation instead of two.
__global__ __launch_bounds__(256) void name_cells_sorted_k(u64 nrec, const u32 *perm, const u8 *cells, u8 *cells_sorted, u8 *outlen) {
  const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nrec) return;
  const uint4 c = *reinterpret_cast<const uint4 *>(cells + 16 * (u64)perm[k]);
  *reinterpret_cast<uint4 *>(cells_sorted + 16 * k) = c;
  outlen[k] = (u8)(c.x & 0xFFu);
}
__global__ __launch_bounds__(256) void emit_names_sorted_k(u64 nrec, const u32 *perm, const u8 *cells_sorted, const u64 *store_off,
                                                          const u8 *store, const u64 *name_off, u8 *out) {
  const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nrec) return;
  u8 *dst = out + name_off[k];
  const uint4 c = *reinterpret_cast<const uint4 *>(cells_sorted + 16 * k);
  const u32 n = c.x & 0xFFu;
  if (n <= 15) {
    const u32 w[4] = {c.x, c.y, c.z, c.w};
    for (u32 i = 0; i <= n; i++) dst[i] = (u8)(w[i >> 2] >> (8 * (i & 3)));
    return;
  }
  const u8 *src = store + store_off[perm[k]];
  dst[0] = (u8)n;
  for (u32 i = 0; i < n; i++) dst[1 + i] = src[i];
}

// row gather: out[k] = rows[perm[k]], `width` bytes per row, row stride `stride` in the source.  A thread produces 16
// consecutive output bytes (one aligned 16-byte store; they span at most two rows when width >= 16) and the grid strides
// over the output: 200 M rows of 150 bytes are 3e10 bytes, far beyond the 2^32 threads a launch may have.
__host__ __device__ inline u32 gather_grid(u64 nrec, u32 width) {
  const u64 chunks = (nrec * (u64)width + 15) / 16, blocks = (chunks + 255) / 256;
  return (u32)(blocks < (1u << 22) ? (blocks ? blocks : 1) : (1u << 22));
}
// (Four chunks per thread with all their loads in front of the stores were measured: 5.1 ms against 4.0 at 50 M x 100 --
// the gather is not short of requests in flight, it is at what random 100-byte rows get out of the memory.)
__global__ __launch_bounds__(256) void gather_rows_k(u64 nrec, const u32 *perm, const u8 *rows, u64 stride, u32 width,
                                                    u8 *out) {
  const u64 total = nrec * (u64)width;
  const u64 step = (u64)gridDim.x * blockDim.x * 16;
  const bool words = (width & 3) == 0 && (stride & 3) == 0;
  for (u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 16; i0 < total; i0 += step) {
    u64 k = i0 / width;
    u32 w = (u32)(i0 - k * width);
    const u8 *src = rows + (u64)perm[k] * stride;
    if (words && w + 16 <= width && i0 + 16 <= total) {  // the chunk lies inside one row: one 16-byte load at a 4-byte boundary
      typedef u32 u32x4u __attribute__((ext_vector_type(4), aligned(4)));
      const u32x4u v = *reinterpret_cast<const u32x4u *>(src + w);
      *reinterpret_cast<uint4 *>(out + i0) = make_uint4(v.x, v.y, v.z, v.w);
    } else if (words) {  // rows are whole words: four 4-byte moves
      u32 v[4] = {0, 0, 0, 0};
      for (int j = 0; j < 4; j++) {
        if (i0 + 4 * j >= total) break;
        v[j] = *reinterpret_cast<const u32 *>(src + w);
        w += 4;
        if (w == width) { w = 0; k++; if (k < nrec) src = rows + (u64)perm[k] * stride; }
      }
      if (i0 + 16 <= total) *reinterpret_cast<uint4 *>(out + i0) = make_uint4(v[0], v[1], v[2], v[3]);
      else for (u64 i = i0; i < total; i++) out[i] = (u8)(v[(i - i0) >> 2] >> (8 * ((i - i0) & 3)));
    } else {
      u32 v[4] = {0, 0, 0, 0};
      const int cnt = total - i0 < 16 ? (int)(total - i0) : 16;
      for (int j = 0; j < cnt; j++) {
        v[j >> 2] |= (u32)src[w] << (8 * (j & 3));
        if (++w == width) { w = 0; k++; if (k < nrec) src = rows + (u64)perm[k] * stride; }
      }
      if (cnt == 16) *reinterpret_cast<uint4 *>(out + i0) = make_uint4(v[0], v[1], v[2], v[3]);
      else for (int j = 0; j < cnt; j++) out[i0 + j] = (u8)(v[j >> 2] >> (8 * (j & 3)));
    }
  }
}

__global__ __launch_bounds__(256) void sum_bytes_k(const u8 *v, u64 n, unsigned long long *out /* zeroed */) {
  u64 acc = 0;
  for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (u64)gridDim.x * blockDim.x * 16) {
    if (i + 16 <= n && (((u64)(v + i)) & 15) == 0) {
      const uint4 w = *reinterpret_cast<const uint4 *>(v + i);
      const u32 x[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int k = 0; k < 4; k++) acc += (x[k] & 0xFF) + ((x[k] >> 8) & 0xFF) + ((x[k] >> 16) & 0xFF) + (x[k] >> 24);
    } else {
      for (u64 j = i; j < n && j < i + 16; j++) acc += v[j];
    }
  }
  for (int o = 32; o; o >>= 1) acc += (u64)__shfl_xor((long long)acc, o);
  if (lane_id() == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}
// classic arrays of a piece -> fused rows (the rare pieces that went through the indexed ingest kernels): row r = q' | packed words [| cell]
__global__ __launch_bounds__(256) void fuse_rows_k(u64 nrec, const u8 *q, u32 L, const u8 *cells /* or null */, const u8 *packed, u32 pstride, u32 pwords,
                                                  u8 *frow, u32 rs, u32 cell_off) {
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrec) return;
  u8 *row = frow + r * (u64)rs;
  for (u32 x = 0; x < L; x++) row[x] = q[r * (u64)L + x];
  u32 *c = reinterpret_cast<u32 *>(row + cell_off);
  for (u32 x = 0; x < pwords; x++) c[x] = reinterpret_cast<const u32 *>(packed + r * (u64)pstride)[x];
  if (cells) for (int x = 0; x < 4; x++) c[pwords + x] = reinterpret_cast<const u32 *>(cells + 16 * r)[x];
}
// q' of fused rows as one contiguous array (SCALCE_OUT_QINPUT for callers that want that)
__global__ __launch_bounds__(256) void compact_q_k(u64 nrec, const u8 *frow, u32 rs, u32 L, u8 *out) {
  const u64 total = nrec * (u64)L;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
    const u64 r = i / L;
    out[i] = frow[r * rs + (i - r * L)];
  }
}

// piecewise copy (sharded runs: quality bytes -> their place in a rank's range of the run-wide stream).  Pieces are
// contiguous in src and sorted by piece_src; dst offsets are arbitrary.  A workgroup takes 16 KB of src in 16-byte granules
// of the ADDRESS space (src itself may start anywhere: a rank's own part begins in the middle of its local stream): one
// binary search per workgroup finds the piece its span starts in (pieces are a few hundred KB: the threads walk on from
// there), a granule that lies inside one piece is one aligned 16-byte load and -- by the alignment of its destination --
// one 16-byte store, four 4-byte stores or sixteen bytes.  (Rounds 1-4: eight bytes per thread, a binary search per
// thread and byte stores -- 10 of the 19 ms the block-range step of a 50 M-read shard took at one rank.)
constexpr u32 CP_CHUNKS = 4;  // granules per thread
__global__ __launch_bounds__(256) void copy_pieces_k(const u8 *src, u8 *dst, const u64 *piece_src, const u64 *piece_dst,
                                                    u32 np, u64 total) {
  __shared__ u32 p_first;
  const u32 a = (u32)((uintptr_t)src & 15u);     // src = sal + a, sal 16-byte aligned
  const u8 *sal = src - a;
  const u64 span0 = (u64)blockIdx.x * (256u * 16u * CP_CHUNKS);   // in sal coordinates: logical offset = x - a
  if (threadIdx.x == 0) {
    const u64 i0 = span0 > a ? span0 - a : 0;
    u32 lo = 0, hi = np;  // last piece with piece_src <= i0
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (piece_src[mid] <= i0) lo = mid; else hi = mid;
    }
    p_first = lo;
  }
  __syncthreads();
  u32 p = p_first;
#pragma unroll 1
  for (u32 c = 0; c < CP_CHUNKS; c++) {
    const u64 x0 = span0 + ((u64)c * 256u + threadIdx.x) * 16u;
    if (x0 >= total + a) break;
    const u64 b0 = x0 > a ? x0 - a : 0, b1 = (x0 + 16 - a) < total ? (x0 + 16 - a) : total;  // logical bytes of this granule
    while (p + 1 < np && piece_src[p + 1] <= b0) p++;
    const uint4 v = *reinterpret_cast<const uint4 *>(sal + x0);
    const bool whole = b1 - b0 == 16 && (p + 1 == np || piece_src[p + 1] >= b1);
    if (whole) {
      u8 *d = dst + piece_dst[p] + (b0 - piece_src[p]);
      const u32 al = (u32)((uintptr_t)d & 15u);
      if (al == 0) *reinterpret_cast<uint4 *>(d) = v;
      else if ((al & 3u) == 0) {
        u32 *d4 = reinterpret_cast<u32 *>(d);
        d4[0] = v.x; d4[1] = v.y; d4[2] = v.z; d4[3] = v.w;
      } else {
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) d[k] = (u8)(w[k >> 2] >> (8 * (k & 3)));
      }
    } else {
      u32 q = p;
      for (u64 i = b0; i < b1; i++) {
        while (q + 1 < np && piece_src[q + 1] <= i) q++;
        const u32 k = (u32)(i + a - x0);
        const u32 word = k < 8 ? (k < 4 ? v.x : v.y) : (k < 12 ? v.z : v.w);   // (selects: an indexed array would live in scratch)
        dst[piece_dst[q] + (i - piece_src[q])] = (u8)(word >> (8 * (k & 3)));
      }
    }
  }
}

}  // namespace scalce
