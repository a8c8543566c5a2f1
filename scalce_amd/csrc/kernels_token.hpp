// kernels_token.hpp -- LCE tokenizer: longest core per read against the DFA, and the exact
// resolution of the reference's order-dependent tie-break.
//
// aho_search (/root/reference/reads.cpp:413-429) keeps, among the longest cores found in a
// read, the first one whose bucket currently holds the most reads (strict >), where "currently"
// means all earlier reads of the run (bin_size is cumulative, reads.cpp:246).  Per read the scan
// is independent; only reads that see two different cores of the same maximal length ("tie
// reads") depend on earlier decisions.  The resolution here is a Jacobi fixed point over the
// tie reads: every read's decision is re-evaluated in parallel from prefix counts of the current
// decisions until nothing changes.  By induction on the input order the fixed point is unique
// and equals the sequential result (the earliest undecided tie read only depends on settled
// ones), so the outcome is bit-exact with -T 1.
#pragma once
#include "kernels_ingest.hpp"

namespace scalce {

constexpr u32 kNoOutD = 0xFFFFFFFFu;
constexpr int kLevelShiftD = 25;
constexpr u32 kBucketMaskD = (1u << kLevelShiftD) - 1;

// Top of the automaton staged in LDS: states are numbered in BFS order, so ids < LDS_STATES are
// the shallowest (most visited) ones.  20 bytes per state.
constexpr int TOK_THREADS = 512;  // 16 waves per CU share two 60 KB copies of the hot states (256: 8 waves, 30.7 ms)

struct TokArgs {
  const uint4 *next;    // 4 x u32 per state
  const u32 *outinfo;   // level<<25 | bucket, or kNoOutD
  u32 n_states;
  u32 lds_states;       // how many leading states to stage (0 = none)
  const u8 *packed;
  u64 nrec;
  int L, stride;
  u32 root_bucket;      // bucket id of "no core" (== number of real buckets)
  u32 *tok_bucket;      // first longest core (bucket id), or root_bucket
  u32 *tok_pos;         // bits 0-15: index of the core's last base; bits 16-30: hits at max level; bit 31: tie
  // k-mer tables (tokenize_kmer_k), 52 KB: t7[16384] u16 | bits8[2048] u32 | out8[2048] u32 | rank8[2048] u16
  const u32 *kmer;
  u32 id8_first;        // first state of depth 8 (= n_states when there is none): ids are BFS ranks, so depth >= 8 <=> id >= this
};
constexpr u32 KMER_T7_WORDS = 16384 / 2, KMER_BITS_WORDS = 2048, KMER_WORDS = KMER_T7_WORDS + 2 * KMER_BITS_WORDS + 2048 / 2;

__device__ __forceinline__ u32 base_at(const u8 *row, int i) { return (row[i >> 2] >> (6 - 2 * (i & 3))) & 3u; }

template <bool USE_LDS>
__global__ __launch_bounds__(TOK_THREADS) void tokenize_k(TokArgs a) {
  extern __shared__ uint4 lds_dyn[];
  uint4 *l_next = lds_dyn;
  u32 *l_out = reinterpret_cast<u32 *>(lds_dyn + a.lds_states);
  if (USE_LDS) {
    for (u32 i = threadIdx.x; i < a.lds_states; i += TOK_THREADS) {
      l_next[i] = a.next[i];
      l_out[i] = a.outinfo[i];
    }
    __syncthreads();
  }
  const u64 r = (u64)blockIdx.x * TOK_THREADS + threadIdx.x;
  if (r >= a.nrec) return;
  const u32 *row = reinterpret_cast<const u32 *>(a.packed + r * (u64)a.stride);
  u32 state = 0, best_lv = 0, best_b = a.root_bucket, best_pos = 0, hits = 0, tie = 0;
  const int nw = (a.L + 15) >> 4;
  for (int w = 0; w < nw; w++) {
    const u32 word = row[w];  // byte j of the row = bases 4j..4j+3, first base in bits 7-6
    const int cnt = (a.L - 16 * w) < 16 ? (a.L - 16 * w) : 16;
    for (int k = 0; k < cnt; k++) {
      const u32 c = (word >> (8 * (k >> 2) + 6 - 2 * (k & 3))) & 3u;
      // one 4-byte transition (a 16-byte row per lane costs four times the LDS bank accesses); bit 31 = the target
      // state has an output
      u32 t;
      if (USE_LDS && state < a.lds_states) t = reinterpret_cast<const u32 *>(l_next)[state * 4 + c];
      else t = reinterpret_cast<const u32 *>(a.next)[(u64)state * 4 + c];
      state = t & 0x7FFFFFFFu;
      if (t >> 31) {
        u32 info;
        if (USE_LDS && state < a.lds_states) info = l_out[state]; else info = a.outinfo[state];
        const u32 lv = info >> kLevelShiftD, b = info & kBucketMaskD;
        if (lv > best_lv) {
          best_lv = lv; best_b = b; best_pos = 16 * w + k; hits = 1; tie = 0;
        } else if (lv == best_lv) {
          hits++;
          if (b != best_b) tie = 1;
        }
      }
    }
  }
  a.tok_bucket[r] = best_b;
  a.tok_pos[r] = best_pos | ((hits > 0x7FFF ? 0x7FFFu : hits) << 16) | (tie << 31);
}

// The same walk with most transitions replaced by lookups that do not depend on the previous state.  With thousands
// of cores of 8 and more bases nearly every 7-mer is a trie node: the walk sits at depth 7-8 (tens of thousands of
// states) and tokenize_k finds four of five transitions in L2 -- 260 GB of random sector reads per 50 M reads, which is
// what bounds it.  But the state after a base is the longest suffix of the text that is a trie node, and when the state
// BEFORE the base has depth <= 7 that suffix is at most 8 long: it is the node of the last 8 bases if they form one
// (bits8 / rank8: states of one depth are numbered in lexicographic order) and otherwise the state the last 7 bases
// lead to from the root (t7).  Only from states of depth >= 8 (a fifth of the positions) the transition itself is read.
struct KmerTables {  // views of the 52 KB block in LDS
  const u16 *t7;
  const u32 *bits8, *out8;
  const u16 *rank8;
  u32 id8_first;
  __device__ __forceinline__ void bind(const u32 *tab, u32 id8) {
    t7 = reinterpret_cast<const u16 *>(tab);
    bits8 = tab + KMER_T7_WORDS;
    out8 = bits8 + KMER_BITS_WORDS;
    rank8 = reinterpret_cast<const u16 *>(out8 + KMER_BITS_WORDS);
    id8_first = id8;
  }
  // transition word (state | has-output << 31) for base c at position pos; `code` = the last 8 bases including c
  __device__ __forceinline__ u32 step(const u32 *next, u32 state, u32 c, u32 code, int pos) const {
    if (pos < 7 || state >= id8_first) return next[(u64)state * 4 + c];
    const u32 wd = bits8[code >> 5], bit = code & 31;
    if ((wd >> bit) & 1)
      return (id8_first + rank8[code >> 5] + (u32)__popc(wd & ((1u << bit) - 1))) | (((out8[code >> 5] >> bit) & 1u) << 31);
    const u32 e = t7[code & 0x3FFFu];
    return (e & 0x7FFFu) | ((e >> 15) << 31);
  }
};

// tokenize_kmer_k is bound by instruction issue (44 per base at 50 M x 100 bp): the lanes of a wave diverge over its four
// ways to a transition (global row, 8-mer bit table, rank, 7-mer table) and the compiler walks them one after the other,
// each behind its own s_waitcnt, with the output word of the new state (another global gather) behind that.  Here a base is
// straight-line code: every lane issues ALL lookups of a base at once -- the table lookups do not depend on the state, only
// the choice between them does; lanes that do not need the global row read row 0 (one line for the whole wave) --, the
// bases of a 16-base word are unrolled (codes come from the byte-swapped words with one v_alignbit), and the output word of
// the state reached is asked for at once and consumed a base later by arithmetic that treats "no output" as a word of 0.
// No load sits behind a branch, so the compiler's own s_waitcnt counts them (a first version issued them from inline asm
// behind branches: the register allocator copied a destination register before the load had landed).
// 1024 threads per workgroup: two workgroups = the CU's 32 waves share two copies of the tables.
constexpr int TOKP_THREADS = 1024;
// T7_OUT: some state of depth <= 7 has an output (a core shorter than 8 bases): the 7-mer table's entries then carry a flag
template <bool T7_OUT>
__global__ __launch_bounds__(TOKP_THREADS) void tokenize_kmer_pipe_k(TokArgs a) {
  // LDS: the 7-mer table as it comes (16 384 x u16) and, per 32 8-mers, ONE 16-byte entry {node bits, output bits,
  // first state of depth 8 + nodes in front of the group} -- one read per base instead of three
  __shared__ u32 t7w[KMER_T7_WORDS];
  __shared__ uint4 g8[KMER_BITS_WORDS];
  {
    const u32 *bits8 = a.kmer + KMER_T7_WORDS, *out8 = bits8 + KMER_BITS_WORDS;
    const u16 *rank8 = reinterpret_cast<const u16 *>(out8 + KMER_BITS_WORDS);
    for (u32 i = threadIdx.x; i < KMER_T7_WORDS; i += TOKP_THREADS) t7w[i] = a.kmer[i];
    for (u32 i = threadIdx.x; i < KMER_BITS_WORDS; i += TOKP_THREADS) g8[i] = make_uint4(bits8[i], out8[i], a.id8_first + rank8[i], 0u);
  }
  __syncthreads();
  const u16 *t7 = reinterpret_cast<const u16 *>(t7w);
  const u64 r = (u64)blockIdx.x * TOKP_THREADS + threadIdx.x;
  if (r >= a.nrec) return;
  const u32 *row = reinterpret_cast<const u32 *>(a.packed + r * (u64)a.stride);
  const char *next_b = reinterpret_cast<const char *>(a.next), *out_b = reinterpret_cast<const char *>(a.outinfo);  // (+ 32-bit byte offsets)
  const u32 id8 = a.id8_first;
  // the best core so far as its output word (level << 25 | bucket); an output word of 0 stands for "no output" and goes
  // through the same arithmetic: while no core has been seen it only disturbs `hits` and `tie`, which the first real core
  // resets (levels are lengths, >= 1) and the end clears if there was none
  u32 state = 0, best_lv = 0, best_info = a.root_bucket, best_pos = 0, hits = 0, tie = 0;
  u32 pend_raw = 0;   // what the output table holds for the state reached a base ago (asked for then, looked at now) ...
  bool pend_has = false;  // ... and whether that state has an output at all
  auto settle = [&](u32 info, u32 pos) {
    const u32 lv = info >> kLevelShiftD;
    if (__ballot(lv >= best_lv) == 0) return;  // most outputs are shorter cores than the best one so far: nothing to do
    const bool gt = lv > best_lv, eq = lv == best_lv, other = info != best_info;
    hits = gt ? 1u : hits + (eq ? 1u : 0u);
    tie = gt ? 0u : (tie | ((eq && other) ? 1u : 0u));
    best_pos = gt ? pos : best_pos;
    best_info = gt ? info : best_info;
    best_lv = gt ? lv : best_lv;
  };
  const int nw = (a.L + 15) >> 4;
  u32 prevs = 0;
  for (int w = 0; w < nw; w++) {
    const int cnt = (a.L - 16 * w) < 16 ? (a.L - 16 * w) : 16;
    const u32 curs = __builtin_bswap32(row[w]);  // base 0 of the word in bits 31-30
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k < cnt) {
        const int pos = 16 * w + k;
        const u32 code = __builtin_amdgcn_alignbit(prevs, curs, 30 - 2 * k) & 0xFFFFu;  // the last 8 bases, the oldest in the top bits
        const bool deep = pos < 7 || state >= id8;
        const u32 tg = *reinterpret_cast<const u32 *>(next_b + (deep ? (state << 4) | ((code & 3u) << 2) : 0u));  // (lanes on the table path: row 0, one line for all of them)
        const uint4 g = g8[code >> 5];
        const u32 e = t7[code & 0x3FFFu];
        const u32 bit = code & 31u;
        settle(pend_has ? pend_raw : 0u, (u32)(pos - 1));
        const u32 t8 = (g.z + (u32)__popc(g.x & ((1u << bit) - 1u))) | (((g.y >> bit) & 1u) << 31);
        const u32 t7v = T7_OUT ? (e & 0x7FFFu) | ((e >> 15) << 31) : e;
        const u32 tl = ((g.x >> bit) & 1u) ? t8 : t7v;
        const u32 t = deep ? tg : tl;
        state = t & 0x7FFFFFFFu;
        pend_has = (t >> 31) != 0;
        pend_raw = *reinterpret_cast<const u32 *>(out_b + (pend_has ? state << 2 : 0u));
      }
    }
    prevs = curs;
  }
  settle(pend_has ? pend_raw : 0u, (u32)(a.L - 1));
  if (best_lv == 0) { hits = 0; tie = 0; }
  a.tok_bucket[r] = best_info & kBucketMaskD;
  a.tok_pos[r] = best_pos | ((hits > 0x7FFF ? 0x7FFFu : hits) << 16) | (tie << 31);
}

// second walk, tie reads only: the distinct cores of maximal length in order of first appearance
struct TieArgs {
  const uint4 *next;
  const u32 *outinfo;
  const u8 *packed;
  int L, stride;
  u32 ntie;
  const u32 *tie_read;   // tie index -> read index
  const u32 *tie_off;    // tie index -> first slot in cand_* (capacity = hits at max level)
  const u32 *bucket_level;
  const u32 *tok_bucket;
  u32 *cand_bucket, *cand_pos;
  u32 *tie_ncand;
  u32 lds_states;
  const u32 *kmer;       // k-mer tables as in TokArgs (USE_KMER)
  u32 id8_first;
};

template <bool USE_LDS, bool USE_KMER = false>
__global__ __launch_bounds__(TOK_THREADS) void tie_candidates_k(TieArgs a) {
  // same staging of the shallow (hot) states as tokenize_k: the second walk used to go to global memory for every
  // transition and took half as long as the first walk for a fifth of the reads
  extern __shared__ uint4 lds_dyn[];
  uint4 *l_next = lds_dyn;
  u32 *l_out = reinterpret_cast<u32 *>(lds_dyn + a.lds_states);
  KmerTables km;
  if (USE_KMER) {  // the k-mer tables of tokenize_kmer_k instead of the staged states (53 KB of dynamic LDS)
    u32 *tab = reinterpret_cast<u32 *>(lds_dyn);
    for (u32 i = threadIdx.x; i < KMER_WORDS; i += TOK_THREADS) tab[i] = a.kmer[i];
    __syncthreads();
    km.bind(tab, a.id8_first);
  } else if (USE_LDS) {
    for (u32 i = threadIdx.x; i < a.lds_states; i += TOK_THREADS) {
      l_next[i] = a.next[i];
      l_out[i] = a.outinfo[i];
    }
    __syncthreads();
  }
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.ntie) return;
  const u32 r = a.tie_read[t];
  const u32 off = a.tie_off[t];
  const u32 lvmax = a.bucket_level[a.tok_bucket[r]];
  const u32 *row = reinterpret_cast<const u32 *>(a.packed + (u64)r * a.stride);
  u32 state = 0, k = 0, code = 0;
  const int nw = (a.L + 15) >> 4;
  for (int w = 0; w < nw; w++) {
    const u32 word = row[w];
    const int cnt = (a.L - 16 * w) < 16 ? (a.L - 16 * w) : 16;
    for (int q = 0; q < cnt; q++) {
      const u32 c = (word >> (8 * (q >> 2) + 6 - 2 * (q & 3))) & 3u;
      code = ((code << 2) | c) & 0xFFFFu;
      u32 tr;
      if (USE_KMER) tr = km.step(reinterpret_cast<const u32 *>(a.next), state, c, code, 16 * w + q);
      else if (USE_LDS && state < a.lds_states) tr = reinterpret_cast<const u32 *>(l_next)[state * 4 + c];
      else tr = reinterpret_cast<const u32 *>(a.next)[(u64)state * 4 + c];
      state = tr & 0x7FFFFFFFu;
      u32 info = kNoOutD;
      if (tr >> 31) { if (!USE_KMER && USE_LDS && state < a.lds_states) info = l_out[state]; else info = a.outinfo[state]; }
      if (info != kNoOutD && (info >> kLevelShiftD) == lvmax) {
        const u32 bk = info & kBucketMaskD;
        bool seen = false;
        for (u32 j = 0; j < k; j++) seen |= (a.cand_bucket[off + j] == bk);
        if (!seen) {
          a.cand_bucket[off + k] = bk;
          a.cand_pos[off + k] = (u32)(16 * w + q);
          k++;
        }
      }
    }
  }
  a.tie_ncand[t] = k;
}

// ---- core tables of realistic size: occurrences found from their STARTS, not by walking the automaton (round 4) ----------
// The walk above costs one dependent transition per base, and with a million cores of 12-32 bases (9.9 M states, 158 MB of
// rows) four of five of them are read from L2 or HBM: 2.3 ns per read against 0.32 with the 15 600-core table.  But the
// automaton is not the contract -- what aho_search (reads.cpp:413-429) reports at position i is the LONGEST CORE ENDING
// THERE, and its result only depends on the occurrences of the longest cores found anywhere in the read, in the order of
// their end positions.  Every core is at least K = min(shortest core, 12) bases long, so every occurrence starts with a
// K-mer that is a depth-K node of the trie:
//   * a bitmap over all 4^K K-mers (2 MB at K = 12: it lives in an XCD's L2) says whether a position can start a core --
//     one probe per position, independent of every other probe (no chain of dependent loads);
//   * where it can (6 % of the positions with a million cores), the node is idK + rank(K-mer) -- nodes of one depth are
//     numbered in lexicographic order -- and the walk goes DOWN the trie from there along the read (a transition is a
//     trie edge iff its bit in `child` is set), noting every node at which a core ends (outinfo's level == the depth).
// Occurrences come out by start position; for cores of one length that is the order of their end positions, and only
// the occurrences of the longest length seen matter at the end: the first of them (bucket, last base), how many, whether
// two different cores are among them -- exactly what tokenize_k reports.  CANDS: the second pass over the tie reads
// (tie_candidates_k's contract: distinct cores of the longest length in order of first appearance).
struct AnchorArgs {
  const u32 *next;       // 4 x u32 per state (bit 31: the target has an output)
  const u32 *outinfo;
  const u64 *bits;       // 4^K bits
  const u32 *rank;       // per 64-bit word: set bits in front of it
  const u32 *child;      // bit 4 s + c: transition c of state s is a trie edge
  const uint4 *single;   // per depth-K node (index: state - idK): {length | bucket << 6, suffix lo, suffix hi, 0} of the ONE core below it, or 0
  u32 K, idK;
  const u8 *packed;
  u64 nrec;
  int L, stride;
  u32 root_bucket;
  u32 *tok_bucket, *tok_pos;
  // CANDS
  u32 ntie;
  const u32 *tie_read, *tie_off, *bucket_level;
  u32 *cand_bucket, *cand_pos, *tie_ncand;
};
template <bool CANDS>
__global__ __launch_bounds__(256) void tokenize_anchor_k(AnchorArgs a) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (CANDS ? (u64)a.ntie : a.nrec)) return;
  const u64 r = CANDS ? (u64)a.tie_read[t] : t;
  const u32 *row = reinterpret_cast<const u32 *>(a.packed + r * (u64)a.stride);
  const u8 *rowb = reinterpret_cast<const u8 *>(row);
  const u32 K = a.K, L = (u32)a.L;
  const u32 kmask = K >= 16 ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
  u32 best_lv = 0, best_b = a.root_bucket, best_pos = 0, hits = 0, tie = 0;
  u32 lvmax = 0, off = 0, ncand = 0;
  if (CANDS) { lvmax = a.bucket_level[a.tok_bucket[r]]; off = a.tie_off[t]; }
  auto occurrence = [&](u32 lv, u32 b, u32 end) {
    if (CANDS) {
      if (lv != lvmax) return;
      bool seen = false;
      for (u32 j = 0; j < ncand; j++) seen |= (a.cand_bucket[off + j] == b);
      if (!seen) { a.cand_bucket[off + ncand] = b; a.cand_pos[off + ncand] = end; ncand++; }
    } else if (lv > best_lv) {
      best_lv = lv; best_b = b; best_pos = end; hits = 1; tie = 0;
    } else if (lv == best_lv) {
      hits++;
      if (b != best_b) tie = 1;
    }
  };
  // the K-mer that ends at base i, from the row's bit string (2 bits per base, first base first: words byte-swapped)
  auto kmer_at = [&](u32 i) -> u32 {
    const u32 bitpos = 2 * (i + 1 - K), wi = bitpos >> 5, sh = bitpos & 31u;
    const u64 two = ((u64)__builtin_bswap32(row[wi]) << 32) | __builtin_bswap32(row[wi + 1]);  // (rows are padded by a spare word)
    return (u32)(two >> (64 - sh - 2 * K)) & kmask;
  };
  // A hit of the bitmap costs a chain of dependent loads (word + rank -> node -> its record, or a walk down the trie), and a
  // wave waits for the lane with the most of them: taken word by word as they were found, a read cost the sum over its words
  // of the busiest lane's hits in each.  So the probes of 128 positions go first (a bit per position that can start a core),
  // then every lane works through ITS hits four at a time -- four words and ranks asked for together, then four records --
  // and reports what it finds in the order of the positions, as before.
  u32 code = 0;
  const u32 nw = (L + 15) >> 4;
  for (u32 seg = 0; seg < nw; seg += 8) {
    u32 hm0 = 0, hm1 = 0, hm2 = 0, hm3 = 0;  // hits of positions 128 seg' .. +127, 32 per word
    for (u32 w = seg; w < nw && w < seg + 8; w++) {
      const u32 word = row[w];  // byte j of the row = bases 4j..4j+3, first base in bits 7-6
      const u32 cnt = (L - 16 * w) < 16 ? (L - 16 * w) : 16;
      // sixteen probes of the bitmap, all issued before any is looked at (a probe behind a branch waits for the one before)
      u32 hitmask = 0;
      {
        u64 wd[16];
        u32 sh[16];
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
          const u32 c = (word >> (8 * (k >> 2) + 6 - 2 * (k & 3))) & 3u;
          code = ((code << 2) | c) & kmask;
          const bool on = k < cnt && 16 * w + k + 1 >= K;
          wd[k] = a.bits[on ? code >> 6 : 0u];
          sh[k] = on ? (code & 63u) : 64u;
        }
#pragma unroll
        for (u32 k = 0; k < 16; k++) hitmask |= (sh[k] < 64u ? (u32)((wd[k] >> sh[k]) & 1ull) : 0u) << k;
      }
      const u32 q = (w - seg) >> 1, hs = hitmask << (16 * ((w - seg) & 1u));
      hm0 |= q == 0 ? hs : 0u; hm1 |= q == 1 ? hs : 0u; hm2 |= q == 2 ? hs : 0u; hm3 |= q == 3 ? hs : 0u;
    }
    while (hm0 | hm1 | hm2 | hm3) {
      u32 pi[4], kc[4], rk[4];
      u64 wdk[4];
      bool pv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {  // the next hit of this lane (none: position of the segment's first K-mer, looked at by nobody)
        const u32 q = hm0 ? 0u : hm1 ? 1u : hm2 ? 2u : 3u;
        const u32 m = hm0 ? hm0 : hm1 ? hm1 : hm2 ? hm2 : hm3;
        pv[u] = m != 0;
        pi[u] = pv[u] ? 16 * seg + 32 * q + (u32)__builtin_ctz(m) : K - 1;
        hm0 &= q == 0 ? hm0 - 1 : 0xFFFFFFFFu; hm1 &= q == 1 ? hm1 - 1 : 0xFFFFFFFFu;
        hm2 &= q == 2 ? hm2 - 1 : 0xFFFFFFFFu; hm3 &= (q == 3 && m) ? hm3 - 1 : 0xFFFFFFFFu;
        kc[u] = kmer_at(pi[u]);
        wdk[u] = a.bits[kc[u] >> 6];
        rk[u] = a.rank[kc[u] >> 6];
      }
      uint4 one[4];
      u32 jn[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        jn[u] = rk[u] + (u32)__popcll(wdk[u] & ((1ull << (kc[u] & 63u)) - 1ull));
        one[u] = a.single[pv[u] ? jn[u] : 0u];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (!pv[u]) continue;
        const u32 i = pi[u];                  // a K-mer that is a trie node ends here: a core may start at p = i + 1 - K
        if (one[u].x) {  // ONE core below this K-mer: its bases behind the K-mer against the read's, in one comparison
          const u32 len = one[u].x & 63u, m = len - K;
          if (i + m < L) {
            bool same = true;
            if (m) {
              const u32 bp = 2 * (i + 1), wi = bp >> 5, sh = bp & 31u;
              const u64 hi64 = ((u64)__builtin_bswap32(row[wi]) << 32) | __builtin_bswap32(row[wi + 1]);
              const u64 lo = (u64)__builtin_bswap32(row[wi + 2]);   // (bits past the row's last base are shifted out below)
              const u64 win = sh ? ((hi64 << sh) | (lo >> (32 - sh))) : hi64;   // the 32 bases from base i + 1 on
              same = (win >> (64 - 2 * m)) == (((u64)one[u].z << 32) | one[u].y);
            }
            if (same) occurrence(len, one[u].x >> 6, i + m);
          }
          continue;
        }
        u32 s = a.idK + jn[u];                // several cores below it: down the trie along the read
        u32 d = K, pos = i;                   // depth of s, index of its last base
        bool has_out = true;                  // (the anchor's own output is not known from a transition word: look)
        for (;;) {
          if (has_out) {
            const u32 info = a.outinfo[s];
            if (info != kNoOutD && (info >> kLevelShiftD) == d) occurrence(d, info & kBucketMaskD, pos);
          }
          if (pos + 1 >= L) break;
          const u32 cn = (rowb[(pos + 1) >> 2] >> (6 - 2 * ((pos + 1) & 3))) & 3u;
          const u32 e = 4 * s + cn;
          if (!((a.child[e >> 5] >> (e & 31u)) & 1u)) break;
          const u32 tr = a.next[e];
          s = tr & 0x7FFFFFFFu;
          has_out = (tr >> 31) != 0;
          d++; pos++;
        }
      }
    }
  }
  if (CANDS) a.tie_ncand[t] = ncand;
  else {
    a.tok_bucket[r] = best_b;
    a.tok_pos[r] = best_pos | ((hits > 0x7FFF ? 0x7FFFu : hits) << 16) | (tie << 31);
  }
}

// The second walk in the shape of tokenize_kmer_pipe_k: straight-line code per base, the output word of the state reached
// looked at a base later.  tie_candidates_k<false, true> walks the same reads with a wait per lookup and 24 waves per CU:
// 1.25 us per base and wave, 2.9 ms for the 9 M tie reads of a 50 M-read shard.
template <bool T7_OUT>
__global__ __launch_bounds__(TOKP_THREADS) void tie_candidates_pipe_k(TieArgs a) {
  __shared__ u32 t7w[KMER_T7_WORDS];
  __shared__ uint4 g8[KMER_BITS_WORDS];
  {
    const u32 *bits8 = a.kmer + KMER_T7_WORDS, *out8 = bits8 + KMER_BITS_WORDS;
    const u16 *rank8 = reinterpret_cast<const u16 *>(out8 + KMER_BITS_WORDS);
    for (u32 i = threadIdx.x; i < KMER_T7_WORDS; i += TOKP_THREADS) t7w[i] = a.kmer[i];
    for (u32 i = threadIdx.x; i < KMER_BITS_WORDS; i += TOKP_THREADS) g8[i] = make_uint4(bits8[i], out8[i], a.id8_first + rank8[i], 0u);
  }
  __syncthreads();
  const u16 *t7 = reinterpret_cast<const u16 *>(t7w);
  const u32 t = blockIdx.x * TOKP_THREADS + threadIdx.x;
  if (t >= a.ntie) return;
  const u32 r = a.tie_read[t], off = a.tie_off[t];
  const u32 lvmax = a.bucket_level[a.tok_bucket[r]];
  const u32 *row = reinterpret_cast<const u32 *>(a.packed + (u64)r * a.stride);
  const char *next_b = reinterpret_cast<const char *>(a.next), *out_b = reinterpret_cast<const char *>(a.outinfo);
  const u32 id8 = a.id8_first;
  // the first four candidates stay in registers and are written at the end: a store inside the walk would sit behind a
  // branch, where the compiler cannot count it, and every base would end in a full wait
  u32 state = 0, k = 0, c0 = 0xFFFFFFFFu, c1 = 0xFFFFFFFFu, c2 = 0xFFFFFFFFu, c3 = 0xFFFFFFFFu, p0 = 0, p1 = 0, p2 = 0, p3 = 0;
  u32 pend_raw = 0;
  bool pend_has = false;
  auto settle = [&](u32 info, u32 pos) {  // a core of the longest length: a new candidate unless its bucket is one already
    const bool cand = (info >> kLevelShiftD) == lvmax;  // (lvmax >= 1: a word of 0, "no output", never is)
    const u32 bk = info & kBucketMaskD;
    const bool fresh = cand && bk != c0 && bk != c1 && bk != c2 && bk != c3;
    if (__ballot(fresh && k >= 4) != 0) {  // a fifth candidate somewhere in the wave: rare
      if (fresh && k >= 4) {
        bool seen = false;
        for (u32 j = 4; j < k && !seen; j++) seen = a.cand_bucket[off + j] == bk;
        if (!seen) { a.cand_bucket[off + k] = bk; a.cand_pos[off + k] = pos; k++; }
      }
    }
    const bool add = fresh && k < 4;
    c0 = (add && k == 0) ? bk : c0; p0 = (add && k == 0) ? pos : p0;
    c1 = (add && k == 1) ? bk : c1; p1 = (add && k == 1) ? pos : p1;
    c2 = (add && k == 2) ? bk : c2; p2 = (add && k == 2) ? pos : p2;
    c3 = (add && k == 3) ? bk : c3; p3 = (add && k == 3) ? pos : p3;
    k += add ? 1u : 0u;
  };
  const int nw = (a.L + 15) >> 4;
  // The row comes in whole, up front (strides are multiples of 16 bytes): a tie read's row is one of 9 M scattered over the
  // array, a wave's 64 rows are 64 different lines, and with a word asked for every sixteen bases the line had left the
  // cache in between -- 720 bytes fetched per tie read for a row of 28 (counters, round 5).
  const uint4 *row4 = reinterpret_cast<const uint4 *>(row);
  const uint4 q0 = row4[0], q1 = a.stride > 16 ? row4[1] : make_uint4(0, 0, 0, 0), q2 = a.stride > 32 ? row4[2] : make_uint4(0, 0, 0, 0);
  auto word = [&](int w) -> u32 {
    u32 v = q0.x;
    v = w == 1 ? q0.y : v; v = w == 2 ? q0.z : v; v = w == 3 ? q0.w : v;
    v = w == 4 ? q1.x : v; v = w == 5 ? q1.y : v; v = w == 6 ? q1.z : v; v = w == 7 ? q1.w : v;
    v = w == 8 ? q2.x : v; v = w == 9 ? q2.y : v; v = w == 10 ? q2.z : v; v = w == 11 ? q2.w : v;
    return v;
  };
  u32 prevs = 0;
  for (int w = 0; w < nw; w++) {
    const int cnt = (a.L - 16 * w) < 16 ? (a.L - 16 * w) : 16;
    const u32 curs = __builtin_bswap32(w < 12 && 4 * w + 4 <= a.stride ? word(w) : row[w]);
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
      if (kk < cnt) {
        const int pos = 16 * w + kk;
        const u32 code = __builtin_amdgcn_alignbit(prevs, curs, 30 - 2 * kk) & 0xFFFFu;
        const bool deep = pos < 7 || state >= id8;
        const u32 tg = *reinterpret_cast<const u32 *>(next_b + (deep ? (state << 4) | ((code & 3u) << 2) : 0u));
        const uint4 g = g8[code >> 5];
        const u32 e = t7[code & 0x3FFFu];
        const u32 bit = code & 31u;
        settle(pend_has ? pend_raw : 0u, (u32)(pos - 1));
        const u32 t8 = (g.z + (u32)__popc(g.x & ((1u << bit) - 1u))) | (((g.y >> bit) & 1u) << 31);
        const u32 t7v = T7_OUT ? (e & 0x7FFFu) | ((e >> 15) << 31) : e;
        const u32 tl = ((g.x >> bit) & 1u) ? t8 : t7v;
        const u32 tt = deep ? tg : tl;
        state = tt & 0x7FFFFFFFu;
        pend_has = (tt >> 31) != 0;
        pend_raw = *reinterpret_cast<const u32 *>(out_b + (pend_has ? state << 2 : 0u));
      }
    }
    prevs = curs;
  }
  settle(pend_has ? pend_raw : 0u, (u32)(a.L - 1));
  if (k > 0) { a.cand_bucket[off] = c0; a.cand_pos[off] = p0; }
  if (k > 1) { a.cand_bucket[off + 1] = c1; a.cand_pos[off + 1] = p1; }
  if (k > 2) { a.cand_bucket[off + 2] = c2; a.cand_pos[off + 2] = p2; }
  if (k > 3) { a.cand_bucket[off + 3] = c3; a.cand_pos[off + 3] = p3; }
  a.tie_ncand[t] = k;
}

// events: one per fixed read (its bucket) and one per (tie read, candidate).  ev_off[r] = first
// event of read r.  Events are created in read order, so a stable sort by bucket leaves every
// bucket's events in input order.
struct EventArgs {
  u64 nrec;
  const u32 *tok_bucket, *tok_pos;
  const u32 *tie_index;  // read -> tie index (valid when tie bit set)
  const u32 *tie_off, *tie_ncand, *cand_bucket;
  const u32 *ev_off;
  u32 *ev_bucket;        // (may be null together with ev_init: the sort on (key, event) pairs carries both in the key)
  u8 *ev_init;           // initial "chosen" flag: fixed reads 1, first candidate 1, others 0
  u32 *ev_key;           // bucket << 2 | candidate of a tie read << 1 | initial flag: what the sort by bucket carries along (32 bits: buckets < 2^30)
};
__global__ __launch_bounds__(256) void events_fill_k(EventArgs a) {
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.nrec) return;
  const u32 e = a.ev_off[r];
  if (a.tok_pos[r] >> 31) {
    const u32 t = a.tie_index[r], off = a.tie_off[t], k = a.tie_ncand[t];
    for (u32 j = 0; j < k; j++) {
      const u32 bk = a.cand_bucket[off + j];
      if (a.ev_bucket) { a.ev_bucket[e + j] = bk; a.ev_init[e + j] = j == 0; }
      if (a.ev_key) a.ev_key[e + j] = (bk << 2) | 2u | (j == 0 ? 1u : 0u);
    }
  } else {
    const u32 bk = a.tok_bucket[r];
    if (a.ev_bucket) { a.ev_bucket[e] = bk; a.ev_init[e] = 1; }
    if (a.ev_key) a.ev_key[e] = (bk << 2) | 1u;
  }
}

struct EvCount {  // events per read
  const u32 *tok_pos, *tie_index, *tie_ncand;
  __device__ u32 operator()(u64 r) const { return (tok_pos[r] >> 31) ? tie_ncand[tie_index[r]] : 1u; }
};
struct TieFlag {
  const u32 *tok_pos;
  __device__ u32 operator()(u64 r) const { return tok_pos[r] >> 31; }
};
struct TieHits {  // candidate capacity of a tie read
  const u32 *tok_pos;
  const u32 *tie_read;
  __device__ u32 operator()(u64 t) const { return (tok_pos[tie_read[t]] >> 16) & 0x7FFFu; }
};
struct TieCompact {  // scatter of the tie-flag scan: read -> tie index and back
  const u32 *tok_pos;
  u32 *tie_index, *tie_read;
  __device__ void operator()(u64 r, u32 idx) const {
    tie_index[r] = idx;
    if (tok_pos[r] >> 31) tie_read[idx] = (u32)r;
  }
};

struct DigitOfArray {  // digit functor for radix_pass: byte `shift/8` of key[payload]
  const u32 *key;
  int shift;
  __device__ u32 operator()(u32 v) const { return (key[v] >> shift) & 255u; }
};

// after the sort: where did each event land, and the initial chosen flags in sorted order
__global__ __launch_bounds__(256) void events_place_k(u32 nev, const u32 *sorted, const u8 *ev_init, u32 *ev_place,
                                                     u8 *chosen) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nev) return;
  const u32 e = sorted[i];
  ev_place[e] = i;
  chosen[i] = ev_init[e];
}

// segment starts of the sorted event array: seg[b] = first position of bucket b (seg[nb] = nev)
__global__ __launch_bounds__(256) void events_segments_k(u32 nev, const u32 *sorted, const u32 *ev_bucket, u32 nb,
                                                        u32 *seg) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nev) return;
  const u32 cur = i < nev ? ev_bucket[sorted[i]] : nb;
  const u32 prev = i ? ev_bucket[sorted[i - 1]] : 0xFFFFFFFFu;
  if (i == 0) {
    for (u32 b = 0; b <= cur && b <= nb; b++) seg[b] = 0;
  } else if (cur != prev) {
    for (u32 b = prev + 1; b <= cur && b <= nb; b++) seg[b] = i;
  }
}

// The sweeps only ever need the events of TIE candidates: a fixed read is in its bucket in every sweep, so what the fixed
// reads in front of a candidate contribute is a constant (fixed_before).  Of the 61 M events of a 50 M-read shard 20 M are
// tie candidates: the flags and prefix sums the sweeps gather from shrink from 246 MB to 80 MB (inside the 256 MB of
// Infinity Cache instead of HBM) and a bucket's rescan walks a third of the events.
//   cidx[p]       compact index of sorted position p (exclusive scan of the tie bit)
//   ev_place[e]   sorted position of event e
struct TieBitOfKey {
  const u32 *keys;
  __device__ u32 operator()(u64 i) const { return (keys[i] >> 1) & 1u; }
};
__global__ __launch_bounds__(256) void events_place_keys_k(u32 nev, const u32 *sorted, const u32 *keys, const u32 *cidx, u32 *ev_place,
                                                          u8 *chosen_t) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nev) return;
  if (keys[i] & 2u) {  // only a tie candidate's place is ever asked for (tie_place_k): a third of the scattered writes
    ev_place[sorted[i]] = i;
    chosen_t[cidx[i]] = (u8)(keys[i] & 1u);
  }
}
// seg[b] = first sorted position of bucket b, seg_t[b] = first compact position, fixed_total[b] = its fixed reads
__global__ __launch_bounds__(256) void events_compact_segments_k(u32 nb1, const u32 *seg, const u32 *cidx, u32 nev, u32 ntev, u32 *seg_t,
                                                                u32 *fixed_total) {
  const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nb1) return;
  const u32 s0 = seg[b], c0 = s0 < nev ? cidx[s0] : ntev;
  seg_t[b] = c0;
  if (b < nb1) {
    const u32 s1 = seg[b + 1], c1 = s1 < nev ? cidx[s1] : ntev;
    fixed_total[b] = (s1 - s0) - (c1 - c0);
  }
}
__global__ __launch_bounds__(256) void events_segments_keys_k(u32 nev, const u32 *keys, u32 nb, u32 *seg) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nev) return;
  const u32 cur = i < nev ? keys[i] >> 2 : nb;
  const u32 prev = i ? keys[i - 1] >> 2 : 0xFFFFFFFFu;
  if (i == 0) {
    for (u32 b = 0; b <= cur && b <= nb; b++) seg[b] = 0;
  } else if (cur != prev) {
    for (u32 b = prev + 1; b <= cur && b <= nb; b++) seg[b] = i;
  }
}

// position of every candidate in the sorted event array, in CSR order (read once per sweep, coalesced)
__global__ __launch_bounds__(256) void tie_place_k(u32 ntie, const u32 *tie_read, const u32 *tie_off, const u32 *tie_ncand,
                                                  const u32 *ev_off, const u32 *ev_place, const u32 *cidx, const u32 *cand_bucket,
                                                  const u32 *seg, const u32 *seg_t, u32 *cand_place, u32 *fixed_before) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntie) return;
  const u32 off = tie_off[t], k = tie_ncand[t], e0 = ev_off[tie_read[t]];
  for (u32 j = 0; j < k; j++) {
    const u32 p = ev_place[e0 + j], c = cidx[p], b = cand_bucket[off + j];
    cand_place[off + j] = c;                                   // compact position of the candidate's event
    fixed_before[off + j] = (p - seg[b]) - (c - seg_t[b]);     // fixed reads of its bucket in front of it
  }
}

struct JacobiArgs {
  u32 ntie;
  const u32 *tie_read, *tie_off, *tie_ncand, *cand_bucket;
  const u32 *cand_place;  // sorted-event position of every candidate (CSR order)
  const u32 *G;         // exclusive prefix of `chosen` over the compact (tie-candidate) events, restarted in every bucket
  const u32 *fixed_before;  // per candidate: fixed reads of its bucket in front of it
  const u64 *prior;     // reads already in each bucket before this shard, or null
  u32 *choice;          // tie index -> chosen candidate ordinal
  u8 *chosen;
  u32 *changed;         // [0] any change
  // per bucket: the first read that a choice change of the previous sweep can affect (= changed read + 1;
  // 0 = all, 0xFFFFFFFF = none).  A decision only depends on counts BEFORE its read, so later changes cannot move it.
  const u32 *dirty_in;
  u32 *dirty_out;       // ... in this sweep (reset by the host before the launch)
  u32 coarse;           // mark whole buckets (see jacobi_k)
};
// (Staging the per-bucket thresholds in LDS with workgroups striding over the ties was measured and is slower, 0.40 against
// 0.33 ms per sweep: the sweeps are bound by the evaluations of the reads that ARE woken up -- three million decisions move
// in the first sweep -- not by looking at the others, which costs 50 us.)
__global__ __launch_bounds__(256) void jacobi_k(JacobiArgs a) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.ntie) return;
  const u32 off = a.tie_off[t], k = a.tie_ncand[t];
  const u32 r = a.tie_read[t];
  {  // a decision can only move if the count of one of its candidate buckets moved before this read
    bool any = false;
    for (u32 j = 0; j < k; j++) any |= a.dirty_in[a.cand_bucket[off + j]] <= r;
    if (!any) return;
  }
  u32 best = 0;
  u64 bestc = 0;
  for (u32 j = 0; j < k; j++) {
    // reads of the bucket in front of this one: earlier shards + fixed reads + tie reads that currently choose it (G is the
    // prefix of the `chosen` flags INSIDE the bucket's segment: one gather per candidate)
    const u64 c = (a.prior ? a.prior[a.cand_bucket[off + j]] : 0ull) + (u64)a.fixed_before[off + j] + (u64)a.G[a.cand_place[off + j]];
    if (j == 0 || c > bestc) {  // strict: an earlier candidate keeps the bucket on equal counts
      best = j;
      bestc = c;
    }
  }
  const u32 old = a.choice[t];
  if (best != old) {
    a.chosen[a.cand_place[off + old]] = 0;
    a.chosen[a.cand_place[off + best]] = 1;
    a.choice[t] = best;
    if (a.coarse) {  // early sweeps, millions of decisions move: "somewhere in this bucket" (a plain store) instead of
                     // "from read r + 1 on" (an atomic minimum on a table of a few thousand hot words)
      a.dirty_out[a.cand_bucket[off + old]] = 0;
      a.dirty_out[a.cand_bucket[off + best]] = 0;
    } else {
      atomicMin(&a.dirty_out[a.cand_bucket[off + old]], r + 1);
      atomicMin(&a.dirty_out[a.cand_bucket[off + best]], r + 1);
    }
    a.changed[0] = 1;  // plain store: every writer stores the same value
  }
}

// ---- the tie-break in windows ------------------------------------------------------------------------------------------
// What the sweeps above cost is not the number of sweeps but what each one looks at: a decision becomes final once every
// earlier decision in its candidate buckets is, so the settled prefix of the input grows by a slice per sweep (1/48 of a
// 50 M-read shard) while every sweep re-evaluates ALL tie reads behind it -- 48 sweeps x 4.5 M evaluations on average, and a
// rescan of the 20 M candidate events' prefix sums each time.  Here the tie reads are taken W at a time in input order:
// the sweeps of a window only evaluate its W reads (everything in front is final, nothing behind is looked at), and only
// the window's own events need new prefix sums.  For that the candidate events are laid out window by window, inside a
// window bucket by bucket in input order ("cell" = window x bucket; the cells of a bucket are consecutive slices of its
// segment of the bucket-sorted events, so a candidate's place is its old place plus one offset per cell).  The `chosen`
// flags are a bitmap over that layout with a count in front of every 64-bit word (P64): reads of a bucket in front of a
// candidate = earlier shards + fixed reads + choices of earlier windows (base) + rank(place) - rank(first place of the
// cell).  A sweep is two launches: tie_window_sweep_k (one thread per tie read of the window) and tie_window_tail_k (one
// workgroup: new counts for the window's few thousand words, or -- nothing moved -- the window's choices folded into
// `base` and on to the next window).  Which window a launch works on is device state: the host enqueues sweeps in batches
// and looks at the state once per batch.  Same fixed point as the global sweeps, reached window by window.
struct TieWinState { u32 window, changed, finished, sweeps; };
constexpr u32 TW_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ u32 tw_rank(const u32 *P64, const u64 *bits, u32 e) {  // chosen flags in front of place e
  const u32 w = e >> 6;
  return P64[w] + (u32)__popcll(bits[w] & ((1ull << (e & 63u)) - 1ull));
}
// Cells.  The candidates of a bucket stand in input order in the bucket-sorted layout, so the candidates of one cell are
// consecutive there: every candidate writes its cell at its place (tw_key_k), one pass over the places finds where cells
// begin and end (tw_heads_k), a scan of the sizes gives the window layout (no atomics: 20 M of them on a few thousand hot
// words took 1.7 ms).
__global__ __launch_bounds__(256) void tw_key_k(u32 ntie, u32 W, u32 nb1, const u32 *tie_off, const u32 *tie_ncand, const u32 *cand_bucket,
                                               const u32 *cand_place, u32 *key_c) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntie) return;
  const u32 off = tie_off[t], k = tie_ncand[t];
  const u32 row = (t / W) * nb1;
  for (u32 j = 0; j < k; j++) key_c[cand_place[off + j]] = row + cand_bucket[off + j];
}
__global__ __launch_bounds__(256) void tw_heads_k(u32 ntev, const u32 *key_c, u32 *cell_first, u32 *cell_end) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ntev) return;
  const u32 cell = key_c[c];
  if (c == 0 || key_c[c - 1] != cell) {
    cell_first[cell] = c;
    if (c) cell_end[key_c[c - 1]] = c;
  }
  if (c == ntev - 1) cell_end[cell] = ntev;
}
struct CellCount {
  const u32 *first, *end;
  __device__ u32 operator()(u64 i) const { return end[i] - first[i]; }
};
// first[cell] -> delta[cell] = first place of the cell in the window layout - its first place in the bucket-sorted layout
__global__ __launch_bounds__(256) void tw_delta_k(u64 ncells, const u32 *cellstart, u32 *first_delta) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncells) first_delta[i] = cellstart[i] - first_delta[i];
}
__global__ __launch_bounds__(256) void tw_cand_k(u32 ntie, u32 W, u32 nb1, const u32 *tie_off, const u32 *tie_ncand, const u32 *cand_bucket,
                                                const u32 *cand_place, const u32 *cellstart, const u32 *delta, u32 *wpos, u32 *rs) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntie) return;
  const u32 off = tie_off[t], k = tie_ncand[t];
  const u64 row = (u64)(t / W) * nb1;
  for (u32 j = 0; j < k; j++) {
    const u64 cell = row + cand_bucket[off + j];
    wpos[off + j] = cand_place[off + j] + delta[cell];
    rs[off + j] = cellstart[cell];
  }
}
struct TieWinArgs {
  u32 ntie, W;
  const u32 *tie_off, *tie_ncand, *cand_bucket, *fixed_before, *wpos, *rs;
  const u64 *prior;   // reads already in each bucket before this shard, or null
  const u32 *base;    // tie reads of earlier windows per bucket
  const u32 *P64;
  const u64 *bits;
  u32 *bits32;        // the same words, for the atomics
  u32 *choice;
  TieWinState *st;
};
__global__ __launch_bounds__(256) void tie_window_sweep_k(TieWinArgs a) {
  if (a.st->finished) return;
  const u32 w = a.st->window;
  const u32 local = blockIdx.x * blockDim.x + threadIdx.x;
  const u64 t64 = (u64)w * a.W + local;
  if (local >= a.W || t64 >= a.ntie) return;
  const u32 t = (u32)t64;
  const u32 off = a.tie_off[t], k = a.tie_ncand[t];
  if (k == 0) { a.choice[t] = 0; return; }
  u32 best = 0;
  u64 bestc = 0;
  for (u32 j = 0; j < k; j++) {
    const u32 bk = a.cand_bucket[off + j];
    const u64 c = (a.prior ? a.prior[bk] : 0ull) + (u64)a.fixed_before[off + j] + (u64)a.base[bk] +
                  (u64)(tw_rank(a.P64, a.bits, a.wpos[off + j]) - tw_rank(a.P64, a.bits, a.rs[off + j]));
    if (j == 0 || c > bestc) {  // strict: an earlier candidate keeps the bucket on equal counts
      best = j;
      bestc = c;
    }
  }
  const u32 old = a.choice[t];
  if (best != old) {
    if (old != TW_NONE) { const u32 e = a.wpos[off + old]; atomicAnd(&a.bits32[e >> 5], ~(1u << (e & 31u))); }
    const u32 e = a.wpos[off + best];
    atomicOr(&a.bits32[e >> 5], 1u << (e & 31u));
    a.choice[t] = best;
    a.st->changed = 1;  // plain store: every writer stores the same value
  }
}
// one workgroup behind every sweep
__global__ __launch_bounds__(1024) void tie_window_tail_k(TieWinState *st, u32 nwin, u32 nb1, const u32 *cellstart, const u64 *bits, u32 *P64,
                                                         u32 *base) {
  __shared__ u32 sm[16];
  if (st->finished) return;
  u32 w = st->window;
  const u32 moved = st->changed;
  __syncthreads();  // (everyone has read the state before thread 0 changes it)
  if (threadIdx.x == 0) st->sweeps++;
  if (!moved) {  // the window is settled: its choices count for the windows behind it
    for (u32 b = threadIdx.x; b < nb1; b += 1024) {
      const u64 cell = (u64)w * nb1 + b;
      const u32 cs = cellstart[cell], ce = cellstart[cell + 1];
      if (ce > cs) base[b] += tw_rank(P64, bits, ce) - tw_rank(P64, bits, cs);
    }
    w++;
    if (threadIdx.x == 0) {
      st->window = w;
      if (w >= nwin) st->finished = 1;
    }
    if (w >= nwin) return;
    __syncthreads();
  }
  // counts in front of the words of window w (its places, and the word its end falls into: rank(end) is asked for)
  const u32 w0 = cellstart[(u64)w * nb1] >> 6, w1 = cellstart[(u64)(w + 1) * nb1] >> 6;
  const u32 n = w1 - w0 + 1, per = (n + 1023) / 1024;
  const u32 lo = w0 + threadIdx.x * per, hi = lo + per < w1 + 1 ? lo + per : w1 + 1;
  u32 mine = 0;
  for (u32 i = lo; i < hi; i++) mine += (u32)__popcll(bits[i]);
  u32 tot;
  u32 run = block_exclusive_sum<u32, 16>(mine, &tot, sm);
  for (u32 i = lo; i < hi; i++) { P64[i] = run; run += (u32)__popcll(bits[i]); }
  if (threadIdx.x == 0) st->changed = 0;
}
// One launch per sweep.  tie_window_tail_k is a launch of its own only because what it computes -- the ranks of the window's
// words, and at the end of a window its choices folded into `base` -- must be the same for every workgroup of the next
// sweep.  But it is little: a window's bitmap is a few thousand words.  Here EVERY workgroup of a sweep derives it again
// for itself, into LDS: it reads what the launch before left behind (did anything move? which window?), and
//   * something moved: the same window again -- ranks of its words from the bitmap as it stands now;
//   * nothing moved: the window is settled.  The ranks of ITS words give each of its cells' totals; the workgroup goes on
//     to the next window, whose own bitmap is still empty (nobody has chosen), so a candidate's count is
//     base + the settled window's total of its bucket.  Workgroup 0 writes base + totals to the other copy of `base`.
// No workgroup waits for another; what a launch leaves for the next one (changed flag, window, which copy of `base`) lives
// in slots indexed by the launch number, so that nothing a late workgroup still reads is overwritten by an early one.
struct TieFusedState { u32 changed[3]; u32 window[2]; u32 base_sel[2]; u32 sweeps; };
struct TieFusedArgs {
  TieWinArgs a;          // (a.base, a.P64, a.st unused)
  u32 n;                 // launch number, from 0
  u32 nwin, nb1, lds_words;
  const u32 *cellstart;
  u32 *base2;            // [2][nb1]
  TieFusedState *fs;
};
constexpr int TWF_THREADS = 1024;
__global__ __launch_bounds__(TWF_THREADS) void tie_window_fused_k(TieFusedArgs g) {
  extern __shared__ u64 snap[];  // the words of one window as this workgroup found them, and (behind them) their ranks
  __shared__ u32 sm[16];
  const TieWinArgs &a = g.a;
  const u32 n = g.n, prev = (n + 2) % 3;
  const u32 moved = g.fs->changed[prev];
  const u32 w = g.fs->window[(n + 1) & 1];
  const u32 sel = g.fs->base_sel[(n + 1) & 1];
  const bool first_wg = blockIdx.x == 0;
  if (w >= g.nwin) {  // all windows settled: hand the state on as it is
    if (first_wg && threadIdx.x == 0) { g.fs->window[n & 1] = w; g.fs->base_sel[n & 1] = sel; }
    return;
  }
  if (first_wg && threadIdx.x == 0) { g.fs->changed[(n + 1) % 3] = 0; g.fs->sweeps++; }  // (read by nobody in this launch)
  // Window `w` -- the one to sweep again, or the one that has just settled: a copy of its words, then their ranks.  Both
  // come from ONE reading of the bitmap, so a count taken from them is exact for every cell whose flags are no longer
  // moving, whatever other workgroups do to other cells meanwhile: a tie read whose predecessors are settled is decided
  // right in every sweep, and the settled prefix of a window only grows.
  const u32 w0 = g.cellstart[(u64)w * g.nb1] >> 6, w1 = g.cellstart[(u64)(w + 1) * g.nb1] >> 6;
  const u32 nw = w1 - w0 + 1, per = (nw + TWF_THREADS - 1) / TWF_THREADS;
  u32 *pre = reinterpret_cast<u32 *>(snap + g.lds_words);
  {
    const u32 lo = threadIdx.x * per, hi = lo + per < nw ? lo + per : nw;
    u32 mine = 0;
    for (u32 i = lo; i < hi; i++) {
      const u64 x = a.bits[w0 + i];
      snap[i] = x;
      mine += (u32)__popcll(x);
    }
    u32 tot;
    u32 run = block_exclusive_sum<u32, 16>(mine, &tot, sm);
    for (u32 i = lo; i < hi; i++) { pre[i] = run; run += (u32)__popcll(snap[i]); }
  }
  __syncthreads();
  auto rank = [&](u32 e) -> u32 {  // chosen flags of the window in front of place e
    const u32 wd = (e >> 6) - w0;
    return pre[wd] + (u32)__popcll(snap[wd] & ((1ull << (e & 63u)) - 1ull));
  };
  const u32 *base_in = g.base2 + (u64)sel * g.nb1;
  const bool settled = moved == 0;
  if (settled) {
    {  // the settled window's choices count for every window behind it: every workgroup folds its share of the buckets (the
       // window's words no longer move, so all copies of their ranks agree; one workgroup walking a million buckets made
       // this launch 0.36 ms long)
      u32 *base_out = g.base2 + (u64)(1 - sel) * g.nb1;
      for (u32 b = blockIdx.x * TWF_THREADS + threadIdx.x; b < g.nb1; b += gridDim.x * TWF_THREADS) {
        const u64 cell = (u64)w * g.nb1 + b;
        const u32 cs = g.cellstart[cell], ce = g.cellstart[cell + 1];
        base_out[b] = base_in[b] + (ce > cs ? rank(ce) - rank(cs) : 0u);
      }
    }
    if (first_wg && threadIdx.x == 0) { g.fs->window[n & 1] = w + 1; g.fs->base_sel[n & 1] = 1 - sel; }
    if (w + 1 >= g.nwin) return;
  } else if (first_wg && threadIdx.x == 0) {
    g.fs->window[n & 1] = w;
    g.fs->base_sel[n & 1] = sel;
  }
  const u32 wt = settled ? w + 1 : w;  // the window whose tie reads are evaluated now
  const u32 local = blockIdx.x * TWF_THREADS + threadIdx.x;
  const u64 t64 = (u64)wt * a.W + local;
  if (local >= a.W || t64 >= a.ntie) return;
  const u32 t = (u32)t64;
  const u32 off = a.tie_off[t], k = a.tie_ncand[t];
  if (k == 0) { a.choice[t] = 0; return; }
  u32 best = 0;
  u64 bestc = 0;
  for (u32 j = 0; j < k; j++) {
    const u32 bk = a.cand_bucket[off + j];
    u64 c = (a.prior ? a.prior[bk] : 0ull) + (u64)a.fixed_before[off + j] + (u64)base_in[bk];
    if (settled) {  // first sweep of a window: its own bitmap is empty; the window in front of it is not in `base` yet
      const u64 cell = (u64)w * g.nb1 + bk;
      const u32 cs = g.cellstart[cell], ce = g.cellstart[cell + 1];
      if (ce > cs) c += (u64)(rank(ce) - rank(cs));
    } else {
      c += (u64)(rank(a.wpos[off + j]) - rank(a.rs[off + j]));
    }
    if (j == 0 || c > bestc) {  // strict: an earlier candidate keeps the bucket on equal counts
      best = j;
      bestc = c;
    }
  }
  const u32 old = a.choice[t];
  if (best != old) {
    if (old != TW_NONE) { const u32 e = a.wpos[off + old]; atomicAnd(&a.bits32[e >> 5], ~(1u << (e & 31u))); }
    const u32 e = a.wpos[off + best];
    atomicOr(&a.bits32[e >> 5], 1u << (e & 31u));
    a.choice[t] = best;
    g.fs->changed[n % 3] = 1;  // plain store: every writer stores the same value
  }
}
// words of the largest window (sizes the LDS of tie_window_fused_k)
__global__ __launch_bounds__(256) void tw_maxwin_k(u32 nwin, u32 nb1, const u32 *cellstart, u32 *out /* zeroed */) {
  const u32 w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < nwin) atomicMax(out, (cellstart[(u64)(w + 1) * nb1] >> 6) - (cellstart[(u64)w * nb1] >> 6) + 1);
}
// after the last launch: base of the copy in use -> counts
__global__ __launch_bounds__(256) void twf_counts_k(u32 nb1, const u32 *fixed_total, const u32 *base2, const TieFusedState *fs, u32 last_n,
                                                   u64 *counts) {
  const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb1) counts[b] = (u64)fixed_total[b] + (u64)base2[(u64)fs->base_sel[last_n & 1] * nb1 + b];
}

__global__ __launch_bounds__(256) void tw_counts_k(u32 nb1, const u32 *fixed_total, const u32 *base, u64 *counts) {
  const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb1) counts[b] = (u64)fixed_total[b] + (u64)base[b];
}

// The bounded way out.  A sweep makes at least the earliest undecided tie read final, so ntie + 1 sweeps always suffice --
// and on an adversarial input (two equally long cores alternating in every read, a 50x pile-up of one short sequence)
// nothing better can be promised: O(ntie^2).  After a fixed number of sweeps the host therefore stops sweeping and lets ONE
// wavefront decide the tie reads in input order, which is the reference's own loop (reads.cpp:413-429 with bin_size
// cumulative, :246): count(b) = prior[b] + fixed reads of b in front of the read + tie reads that chose b so far.
// The lanes stage the next 64 tie reads' candidates in LDS (bucket, prior + fixed_before), lane 0 walks them; the
// running counters sit in LDS when the table has few enough buckets (15 601: yes; a million-core table: global memory).
// ~0.1 us per tie read: 1 s for the 9 M tie reads of a 50 M-read shard, against 47 sweeps = 16 ms when sweeping works.
struct TieSeqArgs {
  u32 ntie, nb1;
  const u32 *tie_off, *tie_ncand, *cand_bucket, *fixed_before;
  const u64 *prior;      // may be null
  u32 *choice;
  u32 *tiecount;         // [nb1] zeroed; used when the counters do not fit LDS
  u32 lds_counters;      // 1: counters in LDS
};
constexpr int TIESEQ_K = 6;  // candidates per tie read staged in LDS (more: read from memory on the spot)
__global__ __launch_bounds__(64) void tie_sequential_k(TieSeqArgs a) {
  extern __shared__ u32 dyn[];
  __shared__ u32 sb[64][TIESEQ_K];
  __shared__ u64 sc[64][TIESEQ_K];
  __shared__ u32 sk[64], so[64];
  const int lane = threadIdx.x;
  u32 *cnt = a.lds_counters ? dyn : a.tiecount;
  if (a.lds_counters) for (u32 i = lane; i < a.nb1; i += 64) dyn[i] = 0;
  __syncthreads();
  for (u32 base = 0; base < a.ntie; base += 64) {
    const u32 t = base + lane;
    if (t < a.ntie) {
      const u32 off = a.tie_off[t], k = a.tie_ncand[t];
      sk[lane] = k; so[lane] = off;
      for (u32 j = 0; j < k && j < (u32)TIESEQ_K; j++) {
        const u32 bk = a.cand_bucket[off + j];
        sb[lane][j] = bk;
        sc[lane][j] = (a.prior ? a.prior[bk] : 0ull) + (u64)a.fixed_before[off + j];
      }
    }
    __syncthreads();
    if (lane == 0) {
      const u32 m = a.ntie - base < 64u ? a.ntie - base : 64u;
      for (u32 i = 0; i < m; i++) {
        const u32 k = sk[i], off = so[i];
        u32 best = 0, bestb = 0;
        u64 bestc = 0;
        for (u32 j = 0; j < k; j++) {
          u32 bk;
          u64 c;
          if (j < (u32)TIESEQ_K) { bk = sb[i][j]; c = sc[i][j]; }
          else { bk = a.cand_bucket[off + j]; c = (a.prior ? a.prior[bk] : 0ull) + (u64)a.fixed_before[off + j]; }
          c += cnt[bk];
          if (j == 0 || c > bestc) { best = j; bestc = c; bestb = bk; }  // strict: the earlier candidate keeps the bucket
        }
        cnt[bestb]++;
        a.choice[base + i] = best;
      }
    }
    __syncthreads();
  }
}
// `chosen` flags of the candidates' events from the decisions (the array is zero when this runs)
__global__ __launch_bounds__(256) void chosen_from_choice_k(u32 ntie, const u32 *tie_off, const u32 *choice, const u32 *cand_place, u8 *chosen) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ntie) chosen[cand_place[tie_off[t] + choice[t]]] = 1;
}

// cross-shard prior counts that moved since the last sweep wake up the reads of those buckets
__global__ __launch_bounds__(256) void prior_dirty_k(u32 nb1, const u64 *prior, u64 *seen, u32 *dirty) {
  const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb1) return;
  const u64 p = prior[b];
  if (p != seen[b]) { seen[b] = p; dirty[b] = 0; }  // moved before every read of this shard
}

// final (bucket, end) per read + API view (pattern index in file order, end)
struct FinalizeArgs {
  u64 nrec;
  const u32 *tok_bucket, *tok_pos, *tie_index, *tie_off, *choice, *cand_bucket, *cand_pos;
  const int32_t *bucket_pattern;
  u32 root_bucket;
  u32 *bucket;     // final bucket id per read
  u16 *end;        // end marker: index of the core's last base + 1, 0 = no core (compress.cpp:682,685)
  int32_t *tokens; // 2 per read: pattern (or -1), end
};
__global__ __launch_bounds__(256) void finalize_k(FinalizeArgs a) {
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.nrec) return;
  u32 b = a.tok_bucket[r], pos = a.tok_pos[r] & 0xFFFFu;
  if (a.tok_pos[r] >> 31) {
    const u32 t = a.tie_index[r], j = a.choice[t];
    b = a.cand_bucket[a.tie_off[t] + j];
    pos = a.cand_pos[a.tie_off[t] + j];
  }
  const u32 e = b == a.root_bucket ? 0u : pos + 1;
  a.bucket[r] = b;
  a.end[r] = (u16)e;
  a.tokens[2 * r] = b == a.root_bucket ? -1 : a.bucket_pattern[b];
  a.tokens[2 * r + 1] = (int32_t)e;
}

// After a sweep only the buckets whose `chosen` flags moved (dirty != none) need new prefix sums, and a decision only
// ever looks at differences inside one bucket's segment: one workgroup per bucket redoes the prefix of its own segment
// (G[pos] = Gseg[b] + flags before pos inside the segment) and the bucket's read count, all others return at once.  The
// 61 M-event global scan per sweep (47 sweeps) this replaces cost as much as the sweeps themselves.
__global__ __launch_bounds__(256) void seg_rescan_k(u32 nb1, const u32 *seg, const u32 *dirty, const u8 *chosen, u32 *G,
                                                   const u32 *fixed_total, u64 *counts) {
  __shared__ u32 sm[4];
  const u32 b = blockIdx.x;
  if (b >= nb1 || dirty[b] == 0xFFFFFFFFu) return;
  const u32 start = seg[b], end = seg[b + 1];
  u32 running = 0;
  for (u32 base = start; base < end; base += 256 * 16) {
    const u32 p0 = base + threadIdx.x * 16;
    u32 f[16], mine = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { f[i] = (p0 + i < end) ? (u32)chosen[p0 + i] : 0u; mine += f[i]; }
    u32 tot;
    u32 ex = running + block_exclusive_sum<u32, 4>(mine, &tot, sm);
#pragma unroll
    for (int i = 0; i < 16; i++) { if (p0 + i < end) G[p0 + i] = ex; ex += f[i]; }
    running += tot;
  }
  if (threadIdx.x == 0) counts[b] = (u64)fixed_total[b] + (u64)running;
}

// The same for tables with more buckets than events per bucket (a million cores: three events per bucket, and a million
// workgroups that each find nothing to do took 0.44 ms): a workgroup takes 256 consecutive buckets, a thread walks its own
// bucket's segment when that is short, and the long ones are done by the whole workgroup one after the other.
__global__ __launch_bounds__(256) void seg_rescan_many_k(u32 nb1, const u32 *seg, const u32 *dirty, const u8 *chosen, u32 *G,
                                                        const u32 *fixed_total, u64 *counts) {
  __shared__ u32 sm[4];
  __shared__ u32 longs[256];
  __shared__ u32 nlong;
  if (threadIdx.x == 0) nlong = 0;
  __syncthreads();
  const u32 b = blockIdx.x * 256 + threadIdx.x;
  if (b < nb1 && dirty[b] != 0xFFFFFFFFu) {
    const u32 start = seg[b], end = seg[b + 1];
    if (end - start <= 64) {
      u32 run = 0;
      for (u32 p = start; p < end; p++) { G[p] = run; run += (u32)chosen[p]; }
      counts[b] = (u64)fixed_total[b] + (u64)run;
    } else {
      longs[atomicAdd(&nlong, 1u)] = b;
    }
  }
  __syncthreads();
  const u32 nl = nlong;
  for (u32 li = 0; li < nl; li++) {
    const u32 bb = longs[li];
    const u32 start = seg[bb], end = seg[bb + 1];
    u32 running = 0;
    for (u32 base = start; base < end; base += 256 * 16) {
      const u32 p0 = base + threadIdx.x * 16;
      u32 f[16], mine = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) { f[i] = (p0 + i < end) ? (u32)chosen[p0 + i] : 0u; mine += f[i]; }
      u32 tot;
      u32 ex = running + block_exclusive_sum<u32, 4>(mine, &tot, sm);
#pragma unroll
      for (int i = 0; i < 16; i++) { if (p0 + i < end) G[p0 + i] = ex; ex += f[i]; }
      running += tot;
    }
    if (threadIdx.x == 0) counts[bb] = (u64)fixed_total[bb] + (u64)running;
  }
}

__global__ __launch_bounds__(256) void add_counts_k(u32 n, const u64 *x, const u64 *y, u64 *out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[i] + y[i];
}

}  // namespace scalce
