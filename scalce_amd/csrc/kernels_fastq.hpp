// Decode side, records back to FASTQ text: the per-record body of decompress.cpp:240-366 -- bucket of the
// record, un-rotation of the 2-bit bases around the core (:331-345), N restore from quality 0 (:350-351), name line,
// '+' line, qualities + phred offset.  One wavefront per run of records, one lane per output byte.
#pragma once
#include "prims.hpp"

namespace scalce {

struct FqBucket {       // one non-empty bucket of the .scalcer directory (built by the host walk)
  u64 first;            // index of its first record
  u64 off;              // byte offset of that record in the read stream
  u32 core_len;
  u32 rec_bytes;        // SZ_READ(L - core_len) + metadata bytes
  char core[32];        // the core itself (patterns are at most 32 bases, reads.cpp:358)
};

struct FqArgs {
  const u8 *reads;          // record stream (bucket headers still in place, skipped through FqBucket::off)
  const FqBucket *dir;
  u32 nbuckets;
  u64 nrecords;
  u32 L, sz_meta;           // sz_meta = 0: no end metadata (mate 2)
  const u8 *qual;           // nrecords x L decoded quality symbols
  u32 phred;
  const u8 *names;          // [u8 n][n bytes]... or nullptr: library mode
  const u64 *name_off;      // start of each record's length byte, nrecords + 1 entries
  u32 lib_len;
  char lib[256];
  u32 mate_digit;           // paired: a trailing "/x" gets this character; 0: names as stored
  u8 *out;
  u64 *rec_off;             // optional: start of every record in the text, nrecords + 1 entries
};

// digits of 0 .. K-1 summed: K for the first digit, K - 10^(t-1) more for every t-digit number and beyond
__device__ __forceinline__ u64 fq_digits_below(u64 K) {
  u64 s = K, p = 10;
  for (int t = 2; t <= 20 && K > p; t++, p *= 10) s += K - p;
  return s;
}
__device__ __forceinline__ u32 fq_digits(u64 K) {
  u32 d = 1;
  for (u64 p = 10; K >= p && d < 20; p *= 10) d++;
  return d;
}

constexpr int FQ_RECORDS_PER_WAVE = 32;

__global__ __launch_bounds__(256) void fastq_records_k(FqArgs a) {
  const int lane = lane_id();
  const u64 w = (u64)blockIdx.x * (blockDim.x / 64) + wave_id();
  const u64 k0 = w * FQ_RECORDS_PER_WAVE;
  if (k0 >= a.nrecords) return;
  const u64 k1 = k0 + FQ_RECORDS_PER_WAVE < a.nrecords ? k0 + FQ_RECORDS_PER_WAVE : a.nrecords;
  // bucket of the first record: last entry with first <= k0
  u32 lo = 0, hi = a.nbuckets;
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (a.dir[mid].first <= k0) lo = mid; else hi = mid;
  }
  u32 b = lo;
  const u32 L = a.L;
  for (u64 K = k0; K < k1; K++) {
    while (b + 1 < a.nbuckets && a.dir[b + 1].first <= K) b++;
    const FqBucket &bk = a.dir[b];
    const u32 corlen = bk.core_len;
    const u8 *rec = a.reads + bk.off + (K - bk.first) * (u64)bk.rec_bytes;
    const u32 nbytes = bk.rec_bytes - a.sz_meta;
    u32 end = 0;
    if (a.sz_meta) end = rec[nbytes] | (a.sz_meta == 2 ? ((u32)rec[nbytes + 1] << 8) : 0u);
    // name and position of the record in the text
    u32 n;
    u64 at;
    const u8 *nm = nullptr;
    if (a.names) {
      const u64 no = a.name_off[K];
      n = a.names[no];
      nm = a.names + no + 1;
      at = (no - K) + K * (2ull * L + 6);
    } else {
      n = a.lib_len + 1 + fq_digits(K);
      at = K * (a.lib_len + 2ull * L + 7) + fq_digits_below(K);
    }
    if (a.rec_off && lane == 0) a.rec_off[K] = at;
    const u8 *q = a.qual + K * (u64)L;
    u8 *o = a.out + at;
    const u32 len = n + 2 * L + 6;
    const bool patch = a.names && a.mate_digit && n > 1 && nm[n - 2] == '/';
    for (u32 t = lane; t < len; t += 64) {
      u8 c;
      if (t == 0) c = '@';
      else if (t <= n) {
        const u32 j = t - 1;
        if (a.names) c = (patch && j == n - 1) ? (u8)a.mate_digit : nm[j];
        else if (j < a.lib_len) c = (u8)a.lib[j];
        else if (j == a.lib_len) c = '.';
        else {  // decimal digit of K, most significant first
          u64 v = K;
          for (u32 r = n - 1 - j; r; r--) v /= 10;
          c = (u8)('0' + v % 10);
        }
      } else if (t == n + 1) c = '\n';
      else if (t < n + 2 + L) {
        const u32 i = t - (n + 2);
        if (q[i] == 0) c = 'N';  // decompress.cpp:350-351
        else {
          // stored order: the part behind the core, then the part in front of it (reads.cpp:432-461)
          u32 p;
          bool in_core = false;
          if (end == 0) p = i;
          else if (i < end - corlen) p = (L - end) + i;
          else if (i < end) { in_core = true; p = i - (end - corlen); }
          else p = i - end;
          c = in_core ? (u8)bk.core[p] : (u8)"ACGT"[(rec[p >> 2] >> ((~p & 3) << 1)) & 3];
        }
      } else if (t == n + 2 + L) c = '\n';
      else if (t == n + 3 + L) c = '+';
      else if (t == n + 4 + L) c = '\n';
      else if (t < n + 5 + 2 * L) c = (u8)(q[t - (n + 5 + L)] + a.phred);
      else c = '\n';
      o[t] = c;
    }
    if (a.rec_off && lane == 0 && K + 1 == a.nrecords) a.rec_off[K + 1] = at + len;
  }
}

}  // namespace scalce
