// kernels_ingest.hpp -- FASTQ text in HBM -> per-record structure-of-arrays.
//   index:   newline positions of the text (the f_gets record reader of thread(),
//            /root/reference/compress.cpp:614-666)
//   unpack:  2-bit bases (getval, const.cpp:47-49), q' = map[q]-offset with N -> 0
//            (output_quality, qualities.cpp:183), stored-name length (output_name, names.cpp:55-57)
//   trigram: order-2 context counters over the input-order q' stream (qualities.cpp:185-198)
// All three are streaming, HBM-bound kernels.
#pragma once
#include "prims.hpp"

namespace scalce {

enum DevErrCode : u32 {
  E_NONE = 0,
  E_LINES = 1,      // newline count not a multiple of 4 / text does not end in '\n'
  E_READLEN = 2,    // a sequence or quality line whose length differs from read_length (compress.cpp:628-634)
  E_NAMELEN = 3,    // stored name longer than 255 bytes (names.cpp:57 keeps the length in one byte)
  E_SYMBOL = 4,     // quality symbol >= 80 with the arithmetic coder on (arithmetic.h:47, tables are 80 wide)
  E_ACOVERFLOW = 5, // a coded block outgrew the reference's 10 MiB output buffer (arithmetic.cpp:101)
  E_PAIRS = 6,      // mates have different record counts
  E_INTERNAL = 7
};
struct DevErr {
  u32 code;
  u32 aux;
  u64 where;
};
__device__ __forceinline__ void dev_fail(DevErr *e, u32 code, u64 where, u32 aux = 0) {
  if (atomicCAS(&e->code, 0u, code) == 0u) {
    e->where = where;
    e->aux = aux;
  }
}

// 0x80 in every byte of x that is zero, exact (no borrow artefacts)
__device__ __forceinline__ u32 zero_bytes(u32 x) {
  u32 t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | x | 0x7F7F7F7Fu);
}
__device__ __forceinline__ u32 newline_mask16(uint4 v) {  // bit i = byte i is '\n'
  // the 0x80 flags of a word become 4 mask bits through one dot product with the weights 1, 2, 4, 8 (v_dot4_u32_u8); the
  // second word of a pair takes 16 .. 128 and adds the first one's bits in the same instruction
  const u32 z0 = zero_bytes(v.x ^ 0x0A0A0A0Au) >> 7, z1 = zero_bytes(v.y ^ 0x0A0A0A0Au) >> 7;
  const u32 z2 = zero_bytes(v.z ^ 0x0A0A0A0Au) >> 7, z3 = zero_bytes(v.w ^ 0x0A0A0A0Au) >> 7;
  const u32 lo = __builtin_amdgcn_udot4(z1, 0x80402010u, __builtin_amdgcn_udot4(z0, 0x08040201u, 0u, false), false);
  const u32 hi = __builtin_amdgcn_udot4(z3, 0x80402010u, __builtin_amdgcn_udot4(z2, 0x08040201u, 0u, false), false);
  return lo | (hi << 8);
}

constexpr int IDX_THREADS = 256;
constexpr int IDX_CHUNKS = 4;                            // 16-byte chunks per thread
constexpr int IDX_TILE = IDX_THREADS * IDX_CHUNKS * 16;  // 16 KiB of text per workgroup

__device__ __forceinline__ u64 text_mask64(const u8 *text, u64 n, u64 off) {
  // newline bitmask of text[off, off+64); off is a multiple of 64, text 16-byte aligned
  u64 m = 0;
  if (off + 64 <= n) {
    const uint4 *p = reinterpret_cast<const uint4 *>(text + off);
#pragma unroll
    for (int c = 0; c < IDX_CHUNKS; c++) m |= (u64)newline_mask16(p[c]) << (16 * c);
  } else {
    for (u64 i = off; i < n; i++) m |= (u64)(text[i] == '\n') << (i - off);
  }
  return m;
}

__global__ __launch_bounds__(IDX_THREADS) void index_count_k(const u8 *text, u64 n, u64 *tile_counts) {
  __shared__ u32 sm[4];
  const u64 off = (u64)blockIdx.x * IDX_TILE + (u64)threadIdx.x * 64;
  u32 c = off < n ? (u32)__popcll(text_mask64(text, n, off)) : 0u;
  u32 tot;
  block_exclusive_sum<u32, 4>(c, &tot, sm);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

__global__ __launch_bounds__(IDX_THREADS) void index_write_k(const u8 *text, u64 n, const u64 *tile_base,
                                                            u64 *line_end, u64 max_lines) {
  __shared__ u32 sm[4];
  const u64 off = (u64)blockIdx.x * IDX_TILE + (u64)threadIdx.x * 64;
  u64 m = off < n ? text_mask64(text, n, off) : 0ull;
  u32 tot;
  u64 g = tile_base[blockIdx.x] + block_exclusive_sum<u32, 4>((u32)__popcll(m), &tot, sm);
  while (m) {
    const int b = __ffsll((long long)m) - 1;
    m &= m - 1;
    if (g < max_lines) line_end[g] = off + b;
    g++;
  }
}

// ---- per-record unpack ------------------------------------------------------------------------
struct UnpackArgs {
  const u8 *text;
  u64 nbytes;
  const u64 *line_end;  // 4 per record
  u64 nrec;
  int L, stride, mate, use_names, no_ac;
  u8 *packed;     // nrec * stride, zero padded rows, base 4j..4j+3 in byte j, first base in bits 7-6
  u8 *q;          // nrec rows of L symbols, qstride bytes apart (L: back to back)
  u8 *namelen;    // nrec (mate 0 only)
  u8 *namecell;   // nrec cells of 16 bytes, cellstride apart (mate 0, names on): [length][first 15 characters]
  // One row per read (round 4): q' | a second copy of the packed bases, qstride = the row's size (128 bytes for a 100 bp
  // read: ONE aligned line), packed2 = q + L.  The emit stage then finds a record's q' and bases -- 100 + 25 bytes --
  // behind ONE random line instead of three (the rows the tokenizer and the order stage walk stay where they are: 32-byte
  // rows read in sequence; the 16-byte name cells keep their own array: with them a row would not fit a line).
  u32 qstride, cellstride;
  u8 *packed2;    // or null: the copy of the packed words inside the fused row, qstride apart, ceil(L / 16) words each
  const u8 *qlut;  // 128 bytes: (values[c] - offset) & 255
  int q_affine;    // >= 0: values[c] == c for every c, q' = (c & 127) - q_affine without the table
  DevErr *err;
  u32 *max_namelen;  // longest stored name of the piece when it exceeds a cell (15 characters), else untouched
};

// 2-bit codes of four ASCII bases packed in a little-endian word -> one byte, first base in bits 7-6.
// Only C/c, G/g, T/t map to 1,2,3; every other byte is 0 (getval, const.cpp:47-49).
__device__ __forceinline__ u32 pack4(u32 v) {
  const u32 up = v & 0xDFDFDFDFu;                                // fold case
  const u32 code = ((v >> 1) ^ (v >> 2)) & 0x03030303u;          // A0 C1 G2 T3 on the letters themselves
  u32 ok = zero_bytes((up & 0xFBFBFBFBu) ^ 0x43434343u) | zero_bytes(up ^ 0x54545454u);  // C, G (they differ in bit 2 only) | T
  ok = (ok >> 7) | (ok >> 6);                                     // 0x80 flag -> 0x03 mask per byte (no multiply: quarter rate)
  return __builtin_amdgcn_udot4(code & ok, 0x01041040u, 0u, false);  // first base x 64 + second x 16 + third x 4 + fourth
}

// aligned little-endian word; bytes at or past `n` read as 0
__device__ __forceinline__ u32 load_word(const u8 *t, u64 a, u64 n) {
  if (a + 4 <= n) return *reinterpret_cast<const u32 *>(t + a);
  u32 v = 0;
  for (int k = 0; k < 4; k++)
    if (a + k < n) v |= (u32)t[a + k] << (8 * k);
  return v;
}
// unaligned little-endian 32-bit fetch with a bound
__device__ __forceinline__ u32 load_u32_unaligned(const u8 *t, u64 at, u64 n) {
  const u64 a = at & ~3ull;
  const u32 sh = (u32)(at & 3) * 8;
  const u32 lo = load_word(t, a, n);
  if (!sh) return lo;
  const u32 hi = load_word(t, a + 4, n);
  return (lo >> sh) | (hi << (32 - sh));
}

template <int PART, typename WordAt, typename ByteAt>
__device__ __forceinline__ void unpack_record_at(const UnpackArgs &a, u64 r, u64 ns, u64 p0, u64 p1, u64 p2, u64 p3, const u8 *lut,
                                                 WordAt word, ByteAt byte_at, u8 *qrow, bool qrow_aligned);
// One record: bases -> 2-bit row (global), qualities -> q' (through `qrow`, LDS or global), name length.
// `word(at)` returns the aligned little-endian 32-bit word that contains text byte `at & ~3`.
// PART: 3 = the whole record; 1 = bases + name only, 2 = qualities only (unpack_tiled_k gives a record to two threads of
// different waves: the work of a thread is a long chain of dependent instructions, and with the 40 KB tile per 128
// records only two waves per SIMD were resident to hide it).
template <int PART = 3, typename WordAt, typename ByteAt>
__device__ __forceinline__ void unpack_record(const UnpackArgs &a, u64 r, const u8 *lut, WordAt word, ByteAt byte_at, u8 *qrow,
                                              bool qrow_aligned) {
  unpack_record_at<PART>(a, r, r ? a.line_end[4 * r - 1] + 1 : 0, a.line_end[4 * r], a.line_end[4 * r + 1], a.line_end[4 * r + 2],
                         a.line_end[4 * r + 3], lut, word, byte_at, qrow, qrow_aligned);
}
// ns = where the record's name line starts, p0 .. p3 = the newlines that end its four lines (text offsets)
template <int PART, typename WordAt, typename ByteAt>
__device__ __forceinline__ void unpack_record_at(const UnpackArgs &a, u64 r, u64 ns, u64 p0, u64 p1, u64 p2, u64 p3, const u8 *lut,
                                                 WordAt word, ByteAt byte_at, u8 *qrow, bool qrow_aligned) {
  if (p1 - p0 - 1 != (u64)a.L || p3 - p2 - 1 != (u64)a.L) {
    if (PART & 1) {
      dev_fail(a.err, E_READLEN, r, (u32)(p1 - p0 - 1));
      if (a.mate == 0) {  // the run ends with an error; until the host sees it, later stages must find a well-formed row
        a.namelen[r] = 0;
        if (a.namecell) *reinterpret_cast<uint4 *>(a.namecell + 16 * r) = make_uint4(0, 0, 0, 0);
      }
    }
    return;
  }
  const int L = a.L;
  u8 *prow = a.packed + r * (u64)a.stride;
  const u64 sb = p0 + 1, sq = p2 + 1;
  auto fetch = [&](u64 at) -> u32 {  // unaligned little-endian 32-bit fetch
    const u32 sh = (u32)(at & 3) * 8;
    const u32 lo = word(at);
    if (!sh) return lo;
    return (lo >> sh) | (word(at + 4) << (32 - sh));
  };
  u32 acc = 0;
  int nacc = 0, wout = 0;
  bool bad = false;
  for (int i = 0; i < L; i += 4) {
    u32 vb = fetch(sb + i);
    u32 vq = fetch(sq + i);
    const int rem = L - i;
    if (rem < 4) {  // last partial group: bytes beyond the line are not part of the read
      const u32 keep = (1u << (8 * rem)) - 1;
      vb &= keep;
      vq &= keep;
    }
    if (PART & 1) {
      acc |= pack4(vb) << (8 * nacc);
      if (++nacc == 4) {
        *reinterpret_cast<u32 *>(prow + 4 * wout) = acc;
        wout++;
        acc = 0;
        nacc = 0;
      }
    }
    if (!(PART & 2)) continue;
    // q' (qualities.cpp:183): exactly 'N' forces the offset, i.e. symbol 0
    const u32 isN = (zero_bytes(vb ^ 0x4E4E4E4Eu) >> 7) * 0xFFu;
    u32 qq;
    if (a.q_affine >= 0) {  // four subtractions in one word, no borrow across bytes: every byte of x is < 128, so with
                            // bit 7 set it cannot go below zero (it returns to the result through the xor)
      const u32 x = vq & 0x7F7F7F7Fu, y = (u32)a.q_affine * 0x01010101u;
      qq = ((x | 0x80808080u) - y) ^ 0x80808080u;
    } else {
      qq = (u32)lut[vq & 127] | ((u32)lut[(vq >> 8) & 127] << 8) | ((u32)lut[(vq >> 16) & 127] << 16) |
           ((u32)lut[(vq >> 24) & 127] << 24);
    }
    qq &= ~isN;
    if (rem >= 4) {
      if (qrow_aligned)
        *reinterpret_cast<u32 *>(qrow + i) = qq;
      else {
        qrow[i] = (u8)qq; qrow[i + 1] = (u8)(qq >> 8); qrow[i + 2] = (u8)(qq >> 16); qrow[i + 3] = (u8)(qq >> 24);
      }
    } else {
      for (int k = 0; k < rem; k++) qrow[i + k] = (u8)(qq >> (8 * k));
      qq &= (1u << (8 * rem)) - 1;
    }
    if (!a.no_ac && ((((qq & 0x7F7F7F7Fu) + 0x30303030u) | qq) & 0x80808080u)) bad = true;  // a byte >= 80
  }
  if (bad) dev_fail(a.err, E_SYMBOL, r);
  if (!(PART & 1)) return;
  // flush the partial word and zero the rest of the row
  for (int w = wout; w < a.stride / 4; w++) {
    *reinterpret_cast<u32 *>(prow + 4 * w) = acc;
    acc = 0;
  }
  if (a.mate == 0) {
    // output_name, names.cpp:55-57: characters after '@' up to the first space or the newline
    u32 len = 0;
    if (a.use_names) {
      u64 i = ns + 1;
      while (i < p0 && byte_at(i) != ' ') i++;
      const u64 l = i - (ns + 1);
      if (l > 255 || p0 <= ns) dev_fail(a.err, E_NAMELEN, r);
      len = l > 255 ? 0u : (u32)l;
      if (a.namecell) {
        u32 w[4] = {len, 0, 0, 0};
        for (u32 k = 0; k < 15 && k < len; k++) w[(k + 1) >> 2] |= (u32)byte_at(ns + 1 + k) << (8 * ((k + 1) & 3));
        *reinterpret_cast<uint4 *>(a.namecell + 16 * r) = make_uint4(w[0], w[1], w[2], w[3]);
      }
      if (len > 15 && a.max_namelen) atomicMax(a.max_namelen, len);
    }
    a.namelen[r] = (u8)len;
  }
}

// Names longer than a cell (15 characters) are kept in a byte store in input order, so that the text of a piece is not
// needed once it is ingested: off[r] = where the name of row r starts (exclusive scan of LongNameLen + bytes in use).
struct LongNameLen {
  const u8 *namelen;
  __device__ u64 operator()(u64 r) const { return namelen[r] > 15 ? (u64)namelen[r] : 0ull; }
};
__global__ __launch_bounds__(256) void long_names_k(u64 nrec, const u8 *text, const u64 *line_end, const u8 *namelen, u64 *off,
                                                   u64 store_base, u8 *store) {
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrec) return;
  const u64 at = off[r] + store_base;
  off[r] = at;
  const u32 n = namelen[r];
  if (n <= 15) return;
  const u64 src = (r ? line_end[4 * r - 1] + 1 : 0) + 1;  // behind the '@'
  for (u32 i = 0; i < n; i++) store[at + i] = text[src + i];
}

// text offset at which line number `line` (0-based) begins, from the per-tile newline counts (index_count_k + scan): the
// tile that holds the newline in front of it, then a walk through that tile.  One wavefront; no line index needed.
__global__ __launch_bounds__(64) void line_offset_k(const u8 *text, u64 nbytes, const u64 *tile_base, u32 ntiles, u64 line, u64 *out) {
  if (line == 0) { if (threadIdx.x == 0) *out = 0; return; }
  const u64 want = line - 1;  // the newline with this index (0-based) ends the line in front
  u32 lo = 0, hi = ntiles;    // largest tile t with tile_base[t] <= want
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (tile_base[mid] <= want) lo = mid; else hi = mid;
  }
  u64 seen = tile_base[lo];
  const u64 t0 = (u64)lo * IDX_TILE, t1 = t0 + IDX_TILE < nbytes ? t0 + IDX_TILE : nbytes;
  for (u64 p = t0; p < t1; p += 64) {
    const u64 i = p + threadIdx.x;
    const u64 m = __ballot(i < t1 && text[i] == '\n');
    const u32 c = (u32)__popcll(m);
    if (seen + c > want) {  // the newline is among these 64 bytes: the (want - seen)-th set bit
      u64 mm = m;
      for (u64 k = seen; k < want; k++) mm &= mm - 1;
      if (threadIdx.x == 0) *out = p + (u64)(__ffsll((long long)mm) - 1) + 1;
      return;
    }
    seen += c;
  }
  if (threadIdx.x == 0) *out = nbytes;  // fewer lines than asked for
}

__global__ void last_record_end_k(const u64 *line_end, u64 nrec, u64 *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = nrec ? line_end[4 * nrec - 1] + 1 : 0;
}

// direct form: every thread reads its record straight from global memory (fallback for long reads / huge names)
__global__ __launch_bounds__(256) void unpack_k(UnpackArgs a) {
  __shared__ u8 lut[128];
  if (threadIdx.x < 128) lut[threadIdx.x] = a.qlut[threadIdx.x];
  __syncthreads();
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.nrec) return;
  u8 *qrow = a.q + r * (u64)a.L;
  unpack_record(a, r, lut, [&](u64 at) { return load_word(a.text, at & ~3ull, a.nbytes); },
                [&](u64 at) { return a.text[at]; }, qrow, ((u64)qrow & 3) == 0);
}

// tiled form: a workgroup copies the contiguous text of UNP_RPB records into LDS with 16-byte loads, the
// threads then work out of LDS, and the q' rows leave through LDS as one contiguous, coalesced block.
// The two tiles are sized for the shard's read length at launch (dynamic LDS: text_cap + 32 + q_cap bytes): 40 KB at
// L = 100, so four workgroups share a CU instead of the two that tiles sized for L = 160 allowed.
constexpr int UNP_RPB = 128;
constexpr int UNP_Q_CAP = 20 * 1024;  // UNP_RPB * L must fit: L <= 160
__host__ __device__ inline u32 unp_text_cap(int L) { return (u32)(((UNP_RPB * (2 * L + 20) + 64) + 15) & ~15); }
__host__ __device__ inline u32 unp_q_cap(int L) { return (u32)((UNP_RPB * L + 15) & ~15); }
__global__ __launch_bounds__(2 * UNP_RPB) void unpack_tiled_k(UnpackArgs a) {
  extern __shared__ __attribute__((aligned(16))) u8 unp_lds[];
  const u32 UNP_TEXT_CAP = unp_text_cap(a.L);
  u8 *tile = unp_lds;
  u8 *qt = unp_lds + UNP_TEXT_CAP + 32;
  __shared__ u8 lut[128];
  const int tid = threadIdx.x;
  if (tid < 128) lut[tid] = a.qlut[tid];
  const u64 r0 = (u64)blockIdx.x * UNP_RPB;
  const u64 r1 = r0 + UNP_RPB < a.nrec ? r0 + UNP_RPB : a.nrec;
  const u64 span0 = r0 ? a.line_end[4 * r0 - 1] + 1 : 0;
  const u64 span1 = a.line_end[4 * (r1 - 1) + 3] + 1;
  const u64 a0 = span0 & ~15ull;
  const u64 tbytes = span1 - a0;
  const bool in_lds = tbytes <= (u64)UNP_TEXT_CAP;  // uniform for the workgroup
  if (in_lds) {
    for (u64 i = (u64)tid * 16; i < tbytes + 8; i += (u64)(2 * UNP_RPB) * 16) {  // +8: the funnel fetch may touch the next word
      uint4 v;
      if (a0 + i + 16 <= a.nbytes) v = *reinterpret_cast<const uint4 *>(a.text + a0 + i);
      else {
        u32 w[4] = {0, 0, 0, 0};
        for (int k = 0; k < 16; k++)
          if (a0 + i + k < a.nbytes) w[k >> 2] |= (u32)a.text[a0 + i + k] << (8 * (k & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
      *reinterpret_cast<uint4 *>(tile + i) = v;
    }
  }
  __syncthreads();
  // waves 0, 1: bases and name of record tid; waves 2, 3: the qualities of record tid - UNP_RPB
  const int rec = tid & (UNP_RPB - 1);
  const bool second = tid >= UNP_RPB;
  const u64 r = r0 + rec;
  if (r < r1) {
    u8 *qrow = qt + (size_t)rec * a.L;
    const bool al = ((a.L & 3) == 0);
    auto w_lds = [&](u64 at) { return *reinterpret_cast<const u32 *>(tile + ((at & ~3ull) - a0)); };
    auto b_lds = [&](u64 at) { return tile[at - a0]; };
    auto w_glb = [&](u64 at) { return load_word(a.text, at & ~3ull, a.nbytes); };
    auto b_glb = [&](u64 at) { return a.text[at]; };
    if (in_lds) {
      if (second) unpack_record<2>(a, r, lut, w_lds, b_lds, qrow, al);
      else unpack_record<1>(a, r, lut, w_lds, b_lds, qrow, al);
    } else {
      if (second) unpack_record<2>(a, r, lut, w_glb, b_glb, qrow, al);
      else unpack_record<1>(a, r, lut, w_glb, b_glb, qrow, al);
    }
  }
  __syncthreads();
  // q' rows of this workgroup are one contiguous range of the output
  const u64 qbytes = (r1 - r0) * (u64)a.L;
  u8 *qdst = a.q + r0 * (u64)a.L;
  if ((((u64)qdst) & 15) == 0) {
    for (u64 i = (u64)tid * 16; i < qbytes; i += (u64)(2 * UNP_RPB) * 16) {
      if (i + 16 <= qbytes) *reinterpret_cast<uint4 *>(qdst + i) = *reinterpret_cast<const uint4 *>(qt + i);
      else for (u64 k = i; k < qbytes; k++) qdst[k] = qt[k];
    }
  } else {
    for (u64 i = tid; i < qbytes; i += 2 * UNP_RPB) qdst[i] = qt[i];
  }
}

// ---- one pass over the text behind the count: line index, record structure and unpack fused ------------------------
// index_write_k + unpack_tiled_k read the text twice more and keep a 32-byte line index per record in between.  For reads
// of 16 .. 160 bases a workgroup instead takes a 16 KB tile of text (plus 1 KB: the records that start in the tile end
// there) into LDS, finds the newlines in it, knows from the tile's line base (index_count_k + scan) which of them end name
// lines, and unpacks the records that START in the tile straight from LDS.  Text read twice instead of three times, no
// line index.  A record that does not end inside the overlap sets `slow` and the host falls back to the indexed kernels.
constexpr u32 ING_TILE = IDX_TILE;       // 16 KB: 30 KB of LDS per workgroup, five workgroups (twenty waves) per CU
constexpr u32 ING_OVER = 1024;
constexpr u32 ING_NLMAX = 2048;
constexpr u32 ING_QCAP = ING_TILE / 2 + 2 * 160 + 64;
constexpr int ING_THREADS = 256;
constexpr int ING_HALF = ING_THREADS / 2;
struct IngestArgs {
  UnpackArgs u;              // text, nbytes, nrec (records to take), L, stride, outputs (already offset to the piece's rows)
  const u64 *tile_base;      // newlines in front of every IDX_TILE (exclusive scan of index_count_k)
  u64 *consumed;             // text offset behind the last record taken
  u32 *slow;                 // set when a record does not fit the overlap
};

// The same pass with the unpack spread over all lanes.  ingest_tiles_k gives a record to two threads that walk its 4-byte
// groups one after the other: a 16 KB tile holds ~70 records of 100 bp, so 140 of the 256 threads run 25-step loops while
// the others wait -- three quarters of the kernel's issue slots.  Here one thread per record only looks at the record's
// structure (lines, name); the bases are then packed in units of one output word (16 bases) and the qualities in units of
// four symbols, units dealt to the lanes in order: every lane busy, consecutive lanes write consecutive words of the
// packed rows and of the q' rows (both are back to back in memory), no staging of q' in LDS.  The smallest and largest q'
// symbol of the tile fall out on the way (sym_range_k read all of q' again for them).
struct Ingest2Args {
  IngestArgs i;
  u64 magic_s, magic_w;   // ceil(2^32 / S), ceil(2^32 / W): S = words per packed row, W = 16-symbol units per read
  u32 step_ks, step_rs, step_kw, step_rw;  // 256 / S, 256 % S, 256 / W, 256 % W
  u16 *tile_minmax;       // per tile: min | max << 8 of its q' symbols (255 | 0 << 8: none), or null
};
constexpr u32 ING2_RECMAX = ING_NLMAX / 4 + 4;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4a __attribute__((ext_vector_type(4), aligned(4)));  // sixteen bytes at a 4-byte boundary
__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b) {
  return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ u32 pk_max_u16(u32 a, u32 b) {
  return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ u32 lds_fetch_u32(const u8 *text, u32 at) {  // unaligned little-endian fetch from the tile
  const u32 *p = reinterpret_cast<const u32 *>(text + (at & ~3u));
  return __builtin_amdgcn_alignbyte(p[1], p[0], at & 3u);
}
__global__ __launch_bounds__(ING_THREADS) void ingest_tiles2_k(Ingest2Args g) {
  const IngestArgs &a = g.i;
  __shared__ __attribute__((aligned(16))) u8 text[ING_TILE + ING_OVER + 32];
  __shared__ u16 nl[ING_NLMAX];
  __shared__ u16 rec_sb[ING2_RECMAX], rec_sq[ING2_RECMAX];
  __shared__ u8 lut[128];
  __shared__ u32 sm[ING_THREADS / 64];
  __shared__ u32 s_count[2];
  __shared__ u32 s_mm[2 * (ING_THREADS / 64)];
  const int tid = threadIdx.x;
  const u32 ti = blockIdx.x;                                       // the tile
  const u64 t0 = (u64)ti * ING_TILE;                               // text offset of the tile
  const u64 avail = a.u.nbytes - t0;
  const u32 len = (u32)(avail < ING_TILE + ING_OVER ? avail : ING_TILE + ING_OVER);
  if (tid < 128) lut[tid] = a.u.qlut[tid];
  if (tid == 0) s_count[1] = 0;
  for (u32 i = (u32)tid * 16; i < len + 8; i += ING_THREADS * 16) {
    uint4 v;
    if (t0 + i + 16 <= a.u.nbytes) v = *reinterpret_cast<const uint4 *>(a.u.text + t0 + i);
    else {
      u32 w[4] = {0, 0, 0, 0};
      for (int k = 0; k < 16; k++)
        if (t0 + i + k < a.u.nbytes) w[k >> 2] |= (u32)a.u.text[t0 + i + k] << (8 * (k & 3));
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *reinterpret_cast<uint4 *>(text + i) = v;
  }
  __syncthreads();
  // newline positions, in order: the tile proper (chunk = 64 bytes per thread), then the overlap -- sixteen chunks, all of
  // wave 0's, which scans them by itself (as a second round of the whole workgroup it was a seventh of the kernel's instructions)
  u32 base;
  {
    const u32 off = (u32)tid * 64;
    u64 m = 0;
    if (off < len) {
      const uint4 *p = reinterpret_cast<const uint4 *>(text + off);
#pragma unroll
      for (int c = 0; c < 4; c++) m |= (u64)newline_mask16(p[c]) << (16 * c);
      if (off + 64 > len) m &= (1ull << (len - off)) - 1;
    }
    u32 at = block_exclusive_sum<u32, ING_THREADS / 64>((u32)__popcll(m), &base, sm);
    while (m) {
      const int bpos = __ffsll((long long)m) - 1;
      m &= m - 1;
      if (at < ING_NLMAX) nl[at] = (u16)(off + bpos);
      at++;
    }
  }
  if (wave_id() == 0) {
    const int lane = lane_id();
    const u32 off = ING_TILE + (u32)lane * 64;
    u64 m = 0;
    if (lane < (int)(ING_OVER / 64) && off < len) {
      const uint4 *p = reinterpret_cast<const uint4 *>(text + off);
#pragma unroll
      for (int c = 0; c < 4; c++) m |= (u64)newline_mask16(p[c]) << (16 * c);
      if (off + 64 > len) m &= (1ull << (len - off)) - 1;
    }
    const u32 cnt = (u32)__popcll(m), inc = wave_inclusive_sum(cnt);
    u32 at = base + inc - cnt;
    while (m) {
      const int bpos = __ffsll((long long)m) - 1;
      m &= m - 1;
      if (at < ING_NLMAX) nl[at] = (u16)(off + bpos);
      at++;
    }
    if (lane == 63) s_count[0] = base + inc;
  }
  __syncthreads();
  const u32 count = s_count[0];
  if (count > ING_NLMAX) { if (tid == 0) atomicExch(a.slow, 1u); return; }
  const u64 G0 = a.tile_base[(u64)ti * (ING_TILE / IDX_TILE)];
  const bool starts_line = t0 == 0 || a.u.text[t0 - 1] == '\n';
  const u32 jmin = starts_line ? 0u : 1u;
  const u32 j0 = jmin + (u32)((4 - ((G0 + jmin) & 3)) & 3);         // first name line that starts here
  const u64 rid0 = (G0 + j0) >> 2;
  const u32 nloc = j0 < count ? (count - j0 + 3) / 4 : 0u;         // name lines that END in tile + overlap
  const int L = a.u.L;
  // one thread per record: is it this tile's (it STARTS here and is whole), are its lines as long as they must be, its name
  for (u32 k = (u32)tid; k < nloc; k += ING_THREADS) {
    const u32 j = j0 + 4 * k;
    const u64 rid = rid0 + k;
    bool take = rid < a.u.nrec;
    u32 ns = 0;
    if (take) {
      ns = j ? (u32)nl[j - 1] + 1 : 0u;
      if (j && j - 1 >= count) take = false;
      else if (ns >= ING_TILE) take = false;                        // starts in the next tile: that workgroup's record
    }
    if (take && j + 3 >= count) {                                   // its four lines must end inside tile + overlap
      take = false;
      atomicExch(a.slow, 1u);
    }
    if (!take) continue;
    atomicAdd(&s_count[1], 1u);                                     // (the records taken are the first ones: a prefix of k)
    const u32 p0 = nl[j], p1 = nl[j + 1], p2 = nl[j + 2], p3 = nl[j + 3];
    if (rid + 1 == a.u.nrec) *a.consumed = t0 + p3 + 1;
    if (p1 - p0 - 1 != (u32)L || p3 - p2 - 1 != (u32)L) {
      dev_fail(a.u.err, E_READLEN, rid, p1 - p0 - 1);
      rec_sb[k] = 0xFFFFu;                                          // nothing of it is unpacked
      rec_sq[k] = 0xFFFFu;
      if (a.u.mate == 0) {  // the run ends with an error; until the host sees it, later stages must find a well-formed row
        a.u.namelen[rid] = 0;
        if (a.u.namecell) { u32x4a z; z.x = z.y = z.z = z.w = 0; *reinterpret_cast<u32x4a *>(a.u.namecell + (u64)a.u.cellstride * rid) = z; }
      }
      continue;
    }
    rec_sb[k] = (u16)(p0 + 1);
    rec_sq[k] = (u16)(p2 + 1);
    if (a.u.mate == 0) {
      // output_name, names.cpp:55-57: characters after '@' up to the first space or the newline
      u32 nlen = 0;
      if (a.u.use_names) {
        // the first space of the name line, four characters at a time (the newline at p0 ends the search)
        u32 i = ns + 1;
        for (; i < p0; i += 4) {
          const u32 z = zero_bytes(lds_fetch_u32(text, i) ^ 0x20202020u);
          if (z) { i += (u32)(__ffs((int)z) - 1) >> 3; break; }
        }
        if (i > p0) i = p0;
        const u32 l = i - (ns + 1);
        if (l > 255 || p0 <= ns) dev_fail(a.u.err, E_NAMELEN, rid);
        nlen = l > 255 ? 0u : l;
        if (a.u.namecell) {
          // cell = [length][15 characters]: the 16 bytes from '@' on with the length in place of the '@', zero behind the name
          u32 w[4];
#pragma unroll
          for (int x = 0; x < 4; x++) {
            const int keepb = (int)nlen + 1 - 4 * x;                // bytes of this word that belong to the cell
            const u32 v = lds_fetch_u32(text, ns + 4 * (u32)x);
            w[x] = keepb >= 4 ? v : keepb <= 0 ? 0u : (v & ((1u << (8 * keepb)) - 1));
          }
          w[0] = (w[0] & 0xFFFFFF00u) | nlen;
          u32x4a cv; cv.x = w[0]; cv.y = w[1]; cv.z = w[2]; cv.w = w[3];
          *reinterpret_cast<u32x4a *>(a.u.namecell + (u64)a.u.cellstride * rid) = cv;  // (4-byte aligned inside a fused row)
        }
        if (nlen > 15 && a.u.max_namelen) atomicMax(a.u.max_namelen, nlen);
      }
      a.u.namelen[rid] = (u8)nlen;
    }
  }
  __syncthreads();
  const u32 ntake = s_count[1];
  // Units are dealt to the threads in order, 256 apart: (record, word) of a thread's next unit follow from the last one by
  // additions (step_k, step_r = 256 / n, 256 % n from the host), and both outputs are back to back in memory -- unit u of
  // the tile is word u behind the tile's first row.
  // bases: one unit = one word of a packed row = 16 bases (zero behind the read).  The last unit of a read is packed like
  // the others -- the bytes behind the line (its newline, the '+' line, the first qualities: all inside the record, which ends
  // inside tile + overlap) go through pack4 with it -- and the codes behind the read are masked off afterwards (a path of
  // its own for that unit ran in every wave beside the common one: a wave's 64 units hold eight or nine last ones).
  const u32 wfull = (u32)L / 16, W = ((u32)L + 15) / 16, ntail = (u32)L - 16 * wfull;  // ntail: bases / symbols of a last unit
  {
    const u32 S = (u32)a.u.stride / 4, units = ntake * S;
    u32 tailmask = 0;  // first base of a byte in its bits 7-6 (pack4): byte m keeps its top 2 * (ntail - 4 m) bits
    for (u32 m = 0; m < 4; m++) {
      const int rem = (int)ntail - 4 * (int)m;
      if (rem > 0) tailmask |= (rem >= 4 ? 0xFFu : (0xFF00u >> (2 * rem)) & 0xFFu) << (8 * m);
    }
    u32 k = (u32)(((u64)(u32)tid * g.magic_s) >> 32), w = (u32)tid - k * S;
    u32 *dst = reinterpret_cast<u32 *>(a.u.packed + rid0 * (u64)a.u.stride);
    for (u32 u = (u32)tid; u < units; u += ING_THREADS) {
      const u32 sb = rec_sb[k];
      if (sb != 0xFFFFu) {
        u32 acc = 0;
        if (w < W) {  // sixteen bases: five aligned words of the tile, four bytes of the row
          const u32 at = sb + 16 * w, sh = at & 3u;
          const u32 *p = reinterpret_cast<const u32 *>(text + (at & ~3u));
          const u32 d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
          acc = pack4(__builtin_amdgcn_alignbyte(d1, d0, sh)) | (pack4(__builtin_amdgcn_alignbyte(d2, d1, sh)) << 8) |
                (pack4(__builtin_amdgcn_alignbyte(d3, d2, sh)) << 16) | (pack4(__builtin_amdgcn_alignbyte(d4, d3, sh)) << 24);
          if (w >= wfull) acc &= tailmask;
        }
        dst[u] = acc;
        if (a.u.packed2 && w < W) reinterpret_cast<u32 *>(a.u.packed2 + (rid0 + k) * (u64)a.u.qstride)[w] = acc;
      }
      k += g.step_ks; w += g.step_rs;
      if (w >= S) { w -= S; k++; }
    }
  }
  // qualities: one unit = sixteen symbols (the last unit of a read: what is left, computed like a whole one and cut when it
  // is stored); q' (qualities.cpp:183): exactly 'N' forces the offset, i.e. symbol 0
  // smallest / largest symbol: packed 16-bit min / max order their halves by the HIGH byte, so a word as it is gives the
  // range of its odd bytes and the word shifted up by one byte that of its even bytes -- no masks
  u32 lo_e = 0xFFFFFFFFu, lo_o = 0xFFFFFFFFu, hi_e = 0, hi_o = 0;
  {
    const u32 units = ntake * W;
    const bool al = (L & 3) == 0;
    const u32 aff = (u32)(a.u.q_affine >= 0 ? a.u.q_affine : 0) * 0x01010101u;
    u32 k = (u32)(((u64)(u32)tid * g.magic_w) >> 32), w = (u32)tid - k * W;
    const u32 QS = a.u.qstride;
    u8 *qtile = a.u.q + rid0 * (u64)QS;
    u32 qoff = k * QS + 16 * w;                                     // (a tile's q' rows: far below 2^32 bytes)
    const u32 qstep = g.step_kw * QS + 16 * g.step_rw, qwrap = QS - 16 * W;
    auto quality = [&](u32 vq) -> u32 {
      if (a.u.q_affine >= 0)  // four subtractions in one word (see unpack_record_at)
        return (((vq & 0x7F7F7F7Fu) | 0x80808080u) - aff) ^ 0x80808080u;
      return (u32)lut[vq & 127] | ((u32)lut[(vq >> 8) & 127] << 8) | ((u32)lut[(vq >> 16) & 127] << 16) | ((u32)lut[(vq >> 24) & 127] << 24);
    };
    auto not_n = [&](u32 vb) -> u32 {  // 0xFF in every byte that is not 'N'
      const u32 zN = zero_bytes(vb ^ 0x4E4E4E4Eu);
      return ~(zN | (zN - (zN >> 7)));
    };
    auto range = [&](u32 ql, u32 qh) {  // ql: bytes that are not symbols hold 0xFF; qh: they hold 0
      lo_o = pk_min_u16(lo_o, ql); lo_e = pk_min_u16(lo_e, ql << 8);
      hi_o = pk_max_u16(hi_o, qh); hi_e = pk_max_u16(hi_e, qh << 8);
    };
    for (u32 u = (u32)tid; u < units; u += ING_THREADS) {
      const u32 sb = rec_sb[k];
      if (sb != 0xFFFFu) {
        const u32 sq = rec_sq[k];
        const u32 ab = sb + 16 * w, aq = sq + 16 * w, shb = ab & 3u, shq = aq & 3u;
        const u32 *pb = reinterpret_cast<const u32 *>(text + (ab & ~3u)), *pq = reinterpret_cast<const u32 *>(text + (aq & ~3u));
        const u32 b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3], b4 = pb[4];
        const u32 q0 = pq[0], q1 = pq[1], q2 = pq[2], q3 = pq[3], q4 = pq[4];
        const u32 r0 = quality(__builtin_amdgcn_alignbyte(q1, q0, shq)) & not_n(__builtin_amdgcn_alignbyte(b1, b0, shb));
        const u32 r1 = quality(__builtin_amdgcn_alignbyte(q2, q1, shq)) & not_n(__builtin_amdgcn_alignbyte(b2, b1, shb));
        const u32 r2 = quality(__builtin_amdgcn_alignbyte(q3, q2, shq)) & not_n(__builtin_amdgcn_alignbyte(b3, b2, shb));
        const u32 r3 = quality(__builtin_amdgcn_alignbyte(q4, q3, shq)) & not_n(__builtin_amdgcn_alignbyte(b4, b3, shb));
        u8 *qdst = qtile + qoff;
        if (w < wfull) {
          if (al) {
            u32x4a v;
            v.x = r0; v.y = r1; v.z = r2; v.w = r3;
            *reinterpret_cast<u32x4a *>(qdst) = v;
          } else {
            const u32 rr[4] = {r0, r1, r2, r3};
            for (int x = 0; x < 16; x++) qdst[x] = (u8)(rr[x >> 2] >> (8 * (x & 3)));
          }
          range(r0, r0); range(r1, r1); range(r2, r2); range(r3, r3);
        } else {  // the last unit of a read: ntail symbols (the same for every read: the trip count is the wave's)
          const u32 rr[4] = {r0, r1, r2, r3};
#pragma unroll
          for (int x = 0; x < 4; x++) {
            const int rem = (int)ntail - 4 * x;
            if (rem > 0) {
              const u32 keep = rem < 4 ? (1u << (8 * rem)) - 1 : 0xFFFFFFFFu, qq = rr[x] & keep;
              if (al) *reinterpret_cast<u32 *>(qdst + 4 * x) = qq;
              else for (int y = 0; y < (rem < 4 ? rem : 4); y++) qdst[4 * x + y] = (u8)(qq >> (8 * y));
              range(qq | ~keep, qq);
            }
          }
        }
      }
      k += g.step_kw; w += g.step_rw; qoff += qstep;
      if (w >= W) { w -= W; k++; qoff += qwrap; }
    }
  }
  u32 lo = min(min((lo_o >> 8) & 0xFFu, lo_o >> 24), min((lo_e >> 8) & 0xFFu, lo_e >> 24));
  u32 hi = max(max((hi_o >> 8) & 0xFFu, hi_o >> 24), max((hi_e >> 8) & 0xFFu, hi_e >> 24));
  for (int o = 32; o; o >>= 1) {
    lo = min(lo, (u32)__shfl_xor((int)lo, o));
    hi = max(hi, (u32)__shfl_xor((int)hi, o));
  }
  if (lane_id() == 0) { s_mm[2 * wave_id()] = lo; s_mm[2 * wave_id() + 1] = hi; }
  __syncthreads();  // (also: the q' rows of the tile are written)
  for (int w = 0; w < ING_THREADS / 64; w++) { lo = min(lo, s_mm[2 * w]); hi = max(hi, s_mm[2 * w + 1]); }
  if (tid == 0 && g.tile_minmax) g.tile_minmax[ti] = (u16)(lo | (hi << 8));
  if (hi >= 80 && !a.u.no_ac) {  // a symbol the coder's tables have no row for (arithmetic.h:47): which record was it?
    for (u32 k = (u32)tid; k < ntake; k += ING_THREADS) {
      if (rec_sb[k] == 0xFFFFu) continue;
      const u8 *qsrc = a.u.q + (rid0 + k) * (u64)a.u.qstride;
      bool bad = false;
      for (int x = 0; x < L; x++) bad |= qsrc[x] >= 80;
      if (bad) { dev_fail(a.u.err, E_SYMBOL, rid0 + k); break; }
    }
  }
}
// smallest / largest q' symbol of the piece from the tiles' (what sym_range_k leaves in minmax; preset to 255.., 0)
__global__ __launch_bounds__(256) void tile_minmax_reduce_k(const u16 *tile_minmax, u32 ntiles, u32 *minmax) {
  u32 lo = 255, hi = 0;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < ntiles; i += gridDim.x * blockDim.x) {
    const u32 v = tile_minmax[i];
    lo = min(lo, v & 0xFFu);
    hi = max(hi, v >> 8);
  }
  for (int o = 32; o; o >>= 1) {
    lo = min(lo, (u32)__shfl_xor((int)lo, o));
    hi = max(hi, (u32)__shfl_xor((int)hi, o));
  }
  if (lane_id() == 0) { atomicMin(&minmax[0], lo); atomicMax(&minmax[1], hi); }
}

// ---- trigram counters -------------------------------------------------------------------------
// freq4[(p0*80+p1)*80+s] += 1 for every symbol of the flat input-order stream that has two
// predecessors (cross-read predecessors included: the reference's prev[] is static,
// qualities.cpp:179).  prev0/prev1 = the two symbols before this shard (500 = none).
//
// The 80^3 x u64 table (4 MB) does not fit LDS and hot trigrams would serialise global atomics, so the
// table is built in slices of leading symbols p0: the slice lives in LDS, every workgroup streams the whole
// q' array with 16-byte loads and counts only the trigrams of its slice with LDS atomics, then adds its
// non-zero counters to the global table.  Only the range [lo, lo + A) of symbols that occur (sym_hist_k, plus
// the two carried-in ones) is laid out, A^2 counters per leading symbol, so a 39-symbol alphabet (q' 2..40)
// fits 20 leading symbols into 122 KB and needs 2 streaming passes instead of the 8 that 6400-counter rows
// of the full 80-symbol alphabet took; a full alphabet still works (4 leading symbols per pass, 20 passes).
// 1024-thread workgroups (one per CU: the counters take 122 KB): the kernel is half instruction issue (20 per symbol) and
// half LDS atomics (2.7 per CU and cycle in this form, tools/ubench_lds_atomics.hip), and sixteen waves overlap the two
// where eight did not: 5.5 -> 3.3 ms per 50 M x 100 bp.  (Rounds 1-2 ran 512 threads: a 1024-thread workgroup needs four
// free wave slots on every SIMD of a CU at once and was not placed at all beside the rows coder, whose waves sat on every
// CU; the one-block-per-lane coder keeps its CUs to itself and leaves the others empty between front-stage kernels.)
constexpr int TRI_THREADS = 1024;
constexpr int TRI_CAP = 30500;      // u32 counters in LDS (122 000 B of the CU's 160 KB)
constexpr int TRI_MAX_PASSES = 20;  // A = 80: 4 leading symbols per pass

// smallest and largest q' symbol of the piece: all the layout of the trigram slices needs (a full histogram with an LDS
// atomic per byte ran at a third of this kernel's rate)
__global__ __launch_bounds__(256) void sym_range_k(const u8 *q, u64 n, u32 *minmax /* [0] = min, [1] = max; preset to 255, 0 */) {
  u32 lo = 255, hi = 0;
  const u64 stride = (u64)gridDim.x * blockDim.x * 16;
  for (u64 t = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 16; t < n; t += stride) {
    if (t + 16 <= n) {
      const uint4 v = *reinterpret_cast<const uint4 *>(q + t);
      const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 a = w[k] & 0xFFu, b = (w[k] >> 8) & 0xFFu, c = (w[k] >> 16) & 0xFFu, d = w[k] >> 24;
        lo = min(lo, min(min(a, b), min(c, d)));
        hi = max(hi, max(max(a, b), max(c, d)));
      }
    } else {
      for (u64 i = t; i < n; i++) { lo = min(lo, (u32)q[i]); hi = max(hi, (u32)q[i]); }
    }
  }
  for (int o = 32; o; o >>= 1) {
    lo = min(lo, (u32)__shfl_xor((int)lo, o));
    hi = max(hi, (u32)__shfl_xor((int)hi, o));
  }
  if (lane_id() == 0) { atomicMin(&minmax[0], lo); atomicMax(&minmax[1], hi); }
}

// range[0] = smallest symbol < 80 that occurs, range[1] = A = span of the occurring symbols (0: none)
// The two symbols in front of a piece (qualities.cpp:179: prev[] runs across reads): the tail of the q' rows already
// held when there are any (q_piece points behind them), else what the caller carried in (500 = none).
// (q0 = the first row of the batch, rows of L symbols qstride bytes apart; the piece begins at symbol symbols_before)
__device__ __forceinline__ u32 q_symbol(const u8 *q0, u32 L, u32 qstride, u64 t) {
  const u64 row = t / L;
  return q0[row * qstride + (t - row * L)];
}
__global__ void tri_prev_k(const u8 *q0, u32 L, u32 qstride, u64 symbols_before, u32 carried0, u32 carried1, u32 *prev /*[2]*/) {
  if (threadIdx.x || blockIdx.x) return;
  if (symbols_before >= 2) { prev[0] = q_symbol(q0, L, qstride, symbols_before - 2); prev[1] = q_symbol(q0, L, qstride, symbols_before - 1); }
  else if (symbols_before == 1) { prev[0] = carried1; prev[1] = q_symbol(q0, L, qstride, 0); }
  else { prev[0] = carried0; prev[1] = carried1; }
}

// range[0] = smallest symbol < 80 to lay out, range[1] = A = span (0: none): the symbols of the piece and the two in front;
// range[2] = 1 when no symbol of the piece lies outside them
__global__ void tri_range_k(const u32 *minmax, const u32 *prev, u32 *range) {
  if (threadIdx.x || blockIdx.x) return;
  u32 lo = minmax[0], hi = minmax[1];
  bool any = lo <= hi && lo < 80;
  if (hi >= 80) hi = 79;  // (symbols >= 80 are an input error reported by the ingest stage)
  for (int k = 0; k < 2; k++)
    if (prev[k] < 80) {
      lo = any ? min(lo, prev[k]) : prev[k];
      hi = any ? max(hi, prev[k]) : prev[k];
      any = true;
    }
  range[0] = any ? lo : 0;
  range[1] = any ? hi - lo + 1 : 0;
  range[2] = minmax[1] < 80 ? 1u : 0u;  // every symbol of the piece lies inside the range: the counting loop need not test
}

// Work is handed out in 64 KB tiles per WAVE through a global counter (one per pass), not by a fixed stride: while
// another shard's arithmetic coder is resident, the waves that share a SIMD with one of its chain waves run at a
// fraction of the speed of the others, and with equal shares the whole workgroup -- and the kernel, one workgroup per
// CU -- waited for them (4.4 -> 14 ms per pass).
constexpr u32 TRI_TILE = 64 * 1024;
// q = the piece's first row; rows of L symbols lie qstride bytes apart (qstride == L: the stream is contiguous).  With
// fused rows (qstride > L, both multiples of 4) a unit of 16 symbols is one 16-byte load at a 4-byte boundary when it lies
// inside a row and is put together from two rows when it does not; a thread finds (row, column) of its first unit by one
// division per tile and moves on by additions.
// (passes [pass, pass_end): a full 80-symbol alphabet needs up to TRI_MAX_PASSES of them, the usual 40 symbols one -- the
//  passes behind the second go out as ONE launch that leaves at the first pass past the alphabet, not as eighteen empty ones)
__global__ __launch_bounds__(TRI_THREADS) void trigram_pass_k(const u8 *q, u64 n, const u32 *prev, u32 pass, u32 pass_end,
                                                             const u32 *range, u64 *freq4, unsigned long long *tile_counter,
                                                             u32 L, u32 qstride) {
  const u32 prev0 = prev[0], prev1 = prev[1];
  // Counters are 16-bit fields, two per word: twice the leading symbols per pass (a 39-symbol alphabet in ONE streaming
  // pass, the full 80 in nine instead of twenty).  LDS has 32-bit atomics only, so a field must never carry into its
  // neighbour: bit 15 is a guard -- the add that finds 0x7FFF takes 32768 off the field again and moves them to the
  // global table.  A carry would need 32768 more adds to land on the same field between two consecutive
  // instructions of that thread; tri_check_k compares the table's total with the number of symbols all the same.
  __shared__ u32 tab[TRI_CAP];
  const u32 lo = range[0], A = range[1];
  if (A == 0) return;
  const u32 AA = A * A, width = (2 * TRI_CAP) / AA;  // leading symbols per pass (>= 9)
  const bool fast = range[2] != 0 && width >= A;        // (range[2]: no symbol outside [lo, lo + A), see tri_range_k)
  for (; pass < pass_end; pass++) {
  const u32 d0 = pass * width;
  if (d0 >= A) return;                          // the alphabet is done
  const u32 used = (A - d0 < width ? A - d0 : width) * AA;
  for (u32 i = threadIdx.x; i < (used + 1) / 2; i += TRI_THREADS) tab[i] = 0;
  __syncthreads();
  const u32 first = lo + d0;                    // leading symbols [first, first + width) belong to this pass
  const int lane = lane_id();
  for (;;) {
    u64 tile = 0;
    if (lane == 0) tile = atomicAdd(tile_counter + pass, 1ull);
    tile = ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(tile >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((u32)tile);
    const u64 tbase = tile * TRI_TILE;
    if (tbase >= n) break;
    const u64 tend = tbase + TRI_TILE < n ? tbase + TRI_TILE : n;
    const bool rows = qstride != L;
    u64 row = 0;
    u32 col = 0;
    if (rows) { const u64 t0 = tbase + (u64)lane * 16; row = t0 / L; col = (u32)(t0 - row * L); }
    const u32 adv_r = 1024u / L, adv_c = 1024u - adv_r * L;
  for (u64 t = tbase + (u64)lane * 16; t < tend; t += 64 * 16) {
    u32 a, b;
    u32 w[4];
    int cnt = 16;
    if (!rows) {
    a = t >= 2 ? q[t - 2] : (t == 1 ? prev1 : prev0);
    b = t >= 1 ? q[t - 1] : prev1;
    if (t + 16 <= n) {
      const uint4 v = *reinterpret_cast<const uint4 *>(q + t);
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
      cnt = (int)(n - t);
      w[0] = w[1] = w[2] = w[3] = 0;
      for (int k = 0; k < cnt; k++) w[k >> 2] |= (u32)q[t + k] << (8 * (k & 3));
    }
    } else {
      const u8 *rp = q + row * qstride;
      // the two symbols in front: in this row, or the tail of the row before
      a = col >= 2 ? rp[col - 2] : (t >= 2 ? (rp - qstride)[L + col - 2] : (t == 1 ? prev1 : prev0));
      b = col >= 1 ? rp[col - 1] : (t >= 1 ? (rp - qstride)[L - 1] : prev1);
      if (col + 16 <= L && t + 16 <= n) {
        typedef u32 u32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const u32x4u v = *reinterpret_cast<const u32x4u *>(rp + col);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      } else {
        // the unit runs into the next row (L and the columns are multiples of 4: no word does), or the piece ends inside it
        cnt = t + 16 <= n ? 16 : (int)(n - t);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u32 c = col + 4 * (u32)k;
          const u8 *p = c < L ? rp + c : rp + qstride + (c - L);
          w[k] = 4 * k < cnt ? *reinterpret_cast<const u32 *>(p) : 0u;
        }
      }
      row += adv_r; col += adv_c;
      if (col >= L) { col -= L; row++; }
    }
    if (fast && t != 0 && cnt == 16) {  // (the piece's first unit: its two symbols in front may be "none"; its last: short)
      // every symbol lies in [lo, lo + A) and the whole alphabet is this pass: no range tests, and the index of a trigram
      // follows from the last one's parts -- i = t A + c with t = (a - lo) A + (b - lo), next t = (b - lo) A + c
      // (v_mad_u32_u24 by hand: full rate; the compiler turns every form of this into v_mad_u64_u32, a quarter of it)
      const u32 lo4 = lo * 0x01010101u;
      auto mad24 = [](u32 x, u32 y, u32 z) -> u32 {  // x y + z, x and y below 2^24
        u32 r;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(y), "v"(z));
        return r;
      };
      u32 cp = b - lo, t2 = mad24(a - lo, A, cp);  // cp: the symbol in front, t2 = the two in front as one number
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {  // four adds on their way before the first answer is looked at
        const u32 wv = w[g4] - lo4;      // (no byte below lo: no borrow)
        u32 ii[4], old[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u32 c = (wv >> (8 * k)) & 255u;
          ii[k] = mad24(t2, A, c);
          t2 = mad24(cp, A, c);
          cp = c;
          old[k] = atomicAdd(&tab[ii[k] >> 1], 1u << ((ii[k] & 1u) * 16));
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u32 i = ii[k], sh = (i & 1u) * 16;
          if (((old[k] >> sh) & 0xFFFFu) == 0x7FFFu) {  // the field just reached 2^15
            atomicSub(&tab[i >> 1], 0x8000u << sh);
            const u32 d = i / AA, rem = i - d * AA, bb = rem / A, cc = rem - bb * A;
            atomicAdd(&freq4[((u64)(lo + d) * 80 + lo + bb) * 80 + lo + cc], 32768ull);
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const u32 c = (w[k >> 2] >> (8 * (k & 3))) & 255u;
      const u32 d = a - first, bb = b - lo, cc = c - lo;  // unsigned: anything outside its range fails the test
      if (k < cnt && d < width && bb < A && cc < A) {
        const u32 i = d * AA + bb * A + cc, sh = (i & 1u) * 16;
        const u32 old = atomicAdd(&tab[i >> 1], 1u << sh);
        if (((old >> sh) & 0xFFFFu) == 0x7FFFu) {  // the field just reached 2^15
          atomicSub(&tab[i >> 1], 0x8000u << sh);
          atomicAdd(&freq4[((u64)(first + d) * 80 + lo + bb) * 80 + lo + cc], 32768ull);
        }
      }
      a = b;
      b = c;
    }
  }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < used; i += TRI_THREADS) {
    const u32 v = (tab[i >> 1] >> ((i & 1u) * 16)) & 0xFFFFu;
    if (v) {
      const u32 d = i / AA, rem = i - d * AA, bb = rem / A, cc = rem - bb * A;
      atomicAdd(&freq4[((u64)(first + d) * 80 + lo + bb) * 80 + lo + cc], (u64)v);
    }
  }
  __syncthreads();  // (the table is zeroed again by the next pass)
  }
}

// every symbol that has two predecessors was counted once: the sum of the table says so (see trigram_pass_k)
__global__ __launch_bounds__(256) void tri_check_k(const u64 *freq4, u64 expected, u64 *acc /* zeroed */, u32 *done /* zeroed */,
                                                  DevErr *err) {
  __shared__ u64 part[256];
  u64 sum = 0;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < 512000u; i += gridDim.x * blockDim.x) sum += freq4[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(reinterpret_cast<unsigned long long *>(acc), (unsigned long long)part[0]);
    __threadfence();
    if (atomicAdd(done, 1u) == gridDim.x - 1) {
      const u64 total = atomicAdd(reinterpret_cast<unsigned long long *>(acc), 0ull);
      if (total != expected) dev_fail(err, E_INTERNAL, total, 16);
    }
  }
}

}  // namespace scalce
