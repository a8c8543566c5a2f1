// scalce_hip.hip -- C ABI (include/scalce_hip.h) over the gfx950 kernels.
// Host side of the hot path: owns the device buffers of a shard, sequences the kernels on the
// caller's stream and reads back the handful of counters that size the next launches.
// There is NO CPU fallback: every stage runs on the device or returns an error.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/scalce_hip.h"
#include "automaton.hpp"
#include "kernels_acl.hpp"
#include "kernels_fastq.hpp"

using namespace scalce;

struct scalce_ctx {
  int device = 0;
  std::string err;
  Automaton A;
  bool have_patterns = false;
  uint4 *d_next = nullptr;
  u32 *d_outinfo = nullptr;
  int32_t *d_bucket_pattern = nullptr;
  u32 *d_bucket_level = nullptr;
  u32 *d_kmer = nullptr;      // k-mer tables of tokenize_kmer_k, or null when the core table does not qualify
  bool kmer_t7_out = false;   // some state of depth <= 7 has an output
  u32 id8_first = 0;
  int tok_lds_states = 0;
  u32 *d_simd_load = nullptr;  // per (XCC, SE, SH, CU, SIMD): coder waves resident there (ac_encode_k's role choice)
  // anchor tables of tokenize_anchor_k (core tables too large for the k-mer tables in LDS), or null
  u64 *d_anchor_bits = nullptr;
  u32 *d_anchor_rank = nullptr, *d_child_bits = nullptr;
  uint4 *d_anchor_single = nullptr;  // per depth-K node: the ONE core below it (length, bucket, packed suffix), or 0 = walk
  u32 anchor_K = 0, anchor_idK = 0;
};

static void set_err(scalce_ctx *c, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  c->err = buf;
}

#define HIP_TRY(ctx, expr)                                                               \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      set_err(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SCALCE_ERR_HIP;                                                             \
    }                                                                                    \
  } while (0)

extern "C" int scalce_ctx_create(int device, scalce_ctx **out) {
  if (!out) return SCALCE_ERR_ARG;
  scalce_ctx *c = new scalce_ctx();
  c->device = device;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    // keep the context so the caller can read the message
    set_err(c, "no HIP device %d (found %d): this library has no CPU path", device, n);
    *out = c;
    return SCALCE_ERR_HIP;
  }
  if (hipSetDevice(device) != hipSuccess) {
    set_err(c, "hipSetDevice(%d) failed", device);
    *out = c;
    return SCALCE_ERR_HIP;
  }
  *out = c;
  HIP_TRY(c, hipMalloc(&c->d_simd_load, sizeof(u32) * AC_SIMD_KEYS));
  HIP_TRY(c, hipMemset(c->d_simd_load, 0, sizeof(u32) * AC_SIMD_KEYS));
  return SCALCE_OK;
}

static void free_tables(scalce_ctx *c) {
  if (c->d_next) hipFree(c->d_next);
  if (c->d_outinfo) hipFree(c->d_outinfo);
  if (c->d_bucket_pattern) hipFree(c->d_bucket_pattern);
  if (c->d_bucket_level) hipFree(c->d_bucket_level);
  if (c->d_kmer) hipFree(c->d_kmer);
  c->d_kmer = nullptr;
  if (c->d_anchor_bits) hipFree(c->d_anchor_bits);
  if (c->d_anchor_rank) hipFree(c->d_anchor_rank);
  if (c->d_child_bits) hipFree(c->d_child_bits);
  if (c->d_anchor_single) hipFree(c->d_anchor_single);
  c->d_anchor_single = nullptr;
  c->d_anchor_bits = nullptr; c->d_anchor_rank = nullptr; c->d_child_bits = nullptr; c->anchor_K = 0;
  c->d_next = nullptr; c->d_outinfo = nullptr; c->d_bucket_pattern = nullptr; c->d_bucket_level = nullptr;
}

extern "C" void scalce_ctx_destroy(scalce_ctx *c) {
  if (!c) return;
  free_tables(c);
  if (c->d_simd_load) hipFree(c->d_simd_load);
  delete c;
}
extern "C" const char *scalce_last_error(const scalce_ctx *c) { return c ? c->err.c_str() : "null context"; }

static int upload_tables(scalce_ctx *c) {
  free_tables(c);
  HIP_TRY(c, hipSetDevice(c->device));
  const Automaton &A = c->A;
  HIP_TRY(c, hipMalloc(&c->d_next, sizeof(uint4) * (size_t)A.n_states));
  HIP_TRY(c, hipMalloc(&c->d_outinfo, sizeof(u32) * (size_t)A.n_states));
  HIP_TRY(c, hipMalloc(&c->d_bucket_pattern, sizeof(int32_t) * A.bucket_pattern.size()));
  HIP_TRY(c, hipMalloc(&c->d_bucket_level, sizeof(u32) * A.bucket_level.size()));
  {  // transitions carry, in bit 31, whether the state they lead to ends a core (itself or through a suffix): the
     // tokenizer then looks the output up only where there is one (~1 % of the positions of a read)
    std::vector<u32> nx(A.next.begin(), A.next.end());
    for (auto &t : nx)
      if (A.outinfo[t] != kNoOutD) t |= 0x80000000u;
    HIP_TRY(c, hipMemcpy(c->d_next, nx.data(), sizeof(u32) * nx.size(), hipMemcpyHostToDevice));
  }
  HIP_TRY(c, hipMemcpy(c->d_outinfo, A.outinfo.data(), sizeof(u32) * A.outinfo.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_bucket_pattern, A.bucket_pattern.data(), sizeof(int32_t) * A.bucket_pattern.size(),
                       hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_bucket_level, A.bucket_level.data(), sizeof(u32) * A.bucket_level.size(),
                       hipMemcpyHostToDevice));
  {
    // k-mer tables for tokenize_kmer_k.  Depth of every state = its distance from the root (a transition raises the
    // depth by at most one and the trie path does); the string of a state of depth 8 follows its first discovery.
    const u32 ns = (u32)A.n_states;
    std::vector<int> depth(ns, -1);
    std::vector<u32> code(ns, 0), order;
    order.reserve(ns);
    depth[0] = 0;
    order.push_back(0);
    for (size_t h = 0; h < order.size(); h++) {
      const u32 st = order[h];
      for (u32 ch = 0; ch < 4; ch++) {
        const u32 t = A.next[(size_t)st * 4 + ch];
        if (depth[t] < 0) { depth[t] = depth[st] + 1; code[t] = (code[st] << 2) | ch; order.push_back(t); }
      }
    }
    bool ok = order.size() == ns;
    u32 id8 = ns, n8 = 0;
    for (u32 st = 0; st < ns && ok; st++) {  // ids are BFS ranks: depth must not decrease with the id
      if (st && depth[st] < depth[st - 1]) ok = false;
      if (depth[st] >= 8 && id8 == ns) id8 = st;
      if (depth[st] == 8) n8++;
    }
    if (ok && id8 > 32768) ok = false;  // t7 keeps a state in 15 bits
    std::vector<u32> tab(KMER_WORDS, 0);
    bool t7_out = false;  // a state of depth <= 7 with an output (a core of fewer than 8 bases in the table)
    if (ok) {
      u16 *t7 = reinterpret_cast<u16 *>(tab.data());
      u32 *bits8 = tab.data() + KMER_T7_WORDS, *out8 = bits8 + KMER_BITS_WORDS;
      u16 *rank8 = reinterpret_cast<u16 *>(out8 + KMER_BITS_WORDS);
      for (u32 x = 0; x < 16384 && ok; x++) {
        u32 st = 0;
        for (int j = 0; j < 7; j++) st = A.next[(size_t)st * 4 + ((x >> (12 - 2 * j)) & 3)];
        if (st >= 32768 || st >= id8) ok = false;
        t7[x] = (u16)(st | (A.outinfo[st] != kNoOutD ? 0x8000u : 0u));
        if (A.outinfo[st] != kNoOutD) t7_out = true;
      }
      u32 prev_code = 0;
      for (u32 i = 0; i < n8 && ok; i++) {  // the depth-8 states: ids id8 .. id8 + n8 - 1 in the order of their 8-mers
        const u32 st = id8 + i;
        if (st >= ns || depth[st] != 8 || (i && code[st] <= prev_code)) { ok = false; break; }
        prev_code = code[st];
        bits8[code[st] >> 5] |= 1u << (code[st] & 31);
        if (A.outinfo[st] != kNoOutD) out8[code[st] >> 5] |= 1u << (code[st] & 31);
      }
      u32 run = 0;
      for (u32 wi = 0; wi < KMER_BITS_WORDS && ok; wi++) {
        if (run > 0xFFFF) ok = false;
        rank8[wi] = (u16)run;
        run += (u32)__builtin_popcount(bits8[wi]);
      }
    }
    if (ok) {
      HIP_TRY(c, hipMalloc(&c->d_kmer, sizeof(u32) * KMER_WORDS));
      HIP_TRY(c, hipMemcpy(c->d_kmer, tab.data(), sizeof(u32) * KMER_WORDS, hipMemcpyHostToDevice));
      c->id8_first = id8;
      c->kmer_t7_out = t7_out;
    }
    // Anchor tables (tokenize_anchor_k): a table whose shallow part does not fit the k-mer tables above -- thousands of
    // 8-mers are fine, a million cores of 12-32 bases are not -- is searched from the occurrences' starts instead of by
    // walking the automaton.  K = min(shortest core, 12).
    // (the k-mer tables only shortcut transitions out of states of depth <= 7: with 400 000 states and more most of the walk
    //  is deeper than that, whether the tables could be built or not)
    const bool want = ns > 400000u;
    if (want && order.size() == ns && A.min_level >= 6 && A.n_buckets > 0) {
      const u32 K = (u32)std::min(A.min_level, 12);
      const size_t nbits = (size_t)1 << (2 * K), nwords = (nbits + 63) / 64;
      std::vector<u64> bits(nwords, 0);
      u32 idK = ns;
      bool lex = true;
      u32 prev = 0;
      for (u32 st = 0; st < ns; st++) {
        if (depth[st] != (int)K) continue;
        if (idK == ns) idK = st;
        else if (code[st] <= prev) lex = false;      // (nodes of one depth are numbered in lexicographic order: BFS over ordered children)
        prev = code[st];
        bits[code[st] >> 6] |= 1ull << (code[st] & 63u);
      }
      // ids of depth K must be one contiguous, sorted range
      u32 nK = 0;
      for (u32 st = 0; st < ns; st++) nK += depth[st] == (int)K;
      for (u32 st = idK; st < idK + nK && lex; st++) if (depth[st] != (int)K) lex = false;
      if (lex && idK < ns) {
        std::vector<u32> rank(nwords);
        u32 run = 0;
        for (size_t w = 0; w < nwords; w++) { rank[w] = run; run += (u32)__builtin_popcountll(bits[w]); }
        std::vector<u32> child(((size_t)ns * 4 + 31) / 32, 0);
        for (u32 st = 0; st < ns; st++)
          for (u32 ch = 0; ch < 4; ch++) {
            const u32 t = A.next[(size_t)st * 4 + ch];
            if (depth[t] == depth[st] + 1) child[((size_t)st * 4 + ch) >> 5] |= 1u << (((size_t)st * 4 + ch) & 31);
          }
        // One probe for most anchors (round 5).  Below 97 % of the depth-K nodes of a million-core table hangs exactly ONE core,
        // on a path without branches: for those the walk down the trie (three dependent loads per base, up to 20 bases) is one
        // 16-byte record -- length, bucket, the bases behind the K-mer packed like the K-mer itself -- and one comparison with
        // the read's own bits.  Any other node (branches, a core that is a prefix of another) keeps record 0 and is walked.
        std::vector<uint4> single(nK, make_uint4(0, 0, 0, 0));
        for (u32 j = 0; j < nK; j++) {
          u32 st = idK + j, d = K, cores = 0, bucket = 0, len = 0;
          u64 suf = 0;
          bool simple = true;
          for (;;) {
            const u32 info = A.outinfo[st];
            if (info != kNoOutD && (info >> kLevelShiftD) == d) { cores++; bucket = info & kBucketMaskD; len = d; }
            u32 nch = 0, chv = 0, nxt = 0;
            for (u32 ch = 0; ch < 4; ch++) {
              const u32 t = A.next[(size_t)st * 4 + ch];
              if (depth[t] == depth[st] + 1) { nch++; chv = ch; nxt = t; }
            }
            if (nch == 0) break;
            if (nch > 1 || cores) { simple = false; break; }   // a branch, or a core with more cores below it
            suf = (suf << 2) | chv;
            st = nxt;
            d++;
            if (d > 44) { simple = false; break; }
          }
          if (simple && cores == 1 && len == d && len - K <= 32 && len < 64 && bucket < (1u << 26))
            single[j] = make_uint4(len | (bucket << 6), (u32)suf, (u32)(suf >> 32), 0);
        }
        HIP_TRY(c, hipMalloc(&c->d_anchor_single, sizeof(uint4) * (size_t)nK));
        HIP_TRY(c, hipMemcpy(c->d_anchor_single, single.data(), sizeof(uint4) * (size_t)nK, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMalloc(&c->d_anchor_bits, sizeof(u64) * nwords));
        HIP_TRY(c, hipMalloc(&c->d_anchor_rank, sizeof(u32) * nwords));
        HIP_TRY(c, hipMalloc(&c->d_child_bits, sizeof(u32) * child.size()));
        HIP_TRY(c, hipMemcpy(c->d_anchor_bits, bits.data(), sizeof(u64) * nwords, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_anchor_rank, rank.data(), sizeof(u32) * nwords, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_child_bits, child.data(), sizeof(u32) * child.size(), hipMemcpyHostToDevice));
        c->anchor_K = K;
        c->anchor_idK = idK;
      }
    }
  }
  // stage as many leading (shallow) states as fit in 60 KiB of LDS: 2 workgroups per CU stay resident
  int cap = (60 * 1024) / 20;
  c->tok_lds_states = A.n_states < cap ? A.n_states : cap;
  c->have_patterns = true;
  return SCALCE_OK;
}

extern "C" int scalce_patterns_load_bin(scalce_ctx *c, const void *blob, size_t n) {
  if (!c || !blob) return SCALCE_ERR_ARG;
  if (!c->A.load_bin(blob, n)) { c->err = c->A.error; return SCALCE_ERR_FORMAT; }
  return upload_tables(c);
}
extern "C" int scalce_patterns_load_text(scalce_ctx *c, const char *text, size_t n) {
  if (!c || !text) return SCALCE_ERR_ARG;
  if (!c->A.load_text(text, n)) { c->err = c->A.error; return SCALCE_ERR_FORMAT; }
  return upload_tables(c);
}
extern "C" int scalce_patterns_count(const scalce_ctx *c) { return c ? (int)c->A.patterns.size() : 0; }
extern "C" int scalce_patterns_states(const scalce_ctx *c) { return c ? c->A.n_states : 0; }
extern "C" int scalce_patterns_buckets(const scalce_ctx *c) { return c ? c->A.n_buckets : 0; }
extern "C" int scalce_pattern_length(const scalce_ctx *c, int p) {
  return (c && p >= 0 && p < (int)c->A.patterns.size()) ? (int)c->A.patterns[p].size() : -1;
}
extern "C" const char *scalce_pattern_string(const scalce_ctx *c, int p) {
  return (c && p >= 0 && p < (int)c->A.patterns.size()) ? c->A.patterns[p].c_str() : nullptr;
}

extern "C" int scalce_patterns_describe_host(const void *blob, size_t n, int is_text, int32_t *bucket_pattern_out,
                                            size_t cap, int32_t *n_states, int32_t *n_buckets) {
  // host-only view of the table builder (no device needed): emission order of the buckets
  Automaton A;
  const bool ok = is_text ? A.load_text(static_cast<const char *>(blob), n) : A.load_bin(blob, n);
  if (!ok) return SCALCE_ERR_FORMAT;
  if (n_states) *n_states = A.n_states;
  if (n_buckets) *n_buckets = A.n_buckets;
  if (bucket_pattern_out)
    for (size_t i = 0; i < cap && i < A.bucket_pattern.size(); i++) bucket_pattern_out[i] = A.bucket_pattern[i];
  return SCALCE_OK;
}

extern "C" void scalce_params_default(scalce_params *p) {
  std::memset(p, 0, sizeof *p);
  p->use_names = 1;
  for (int m = 0; m < 2; m++) {
    p->qmap[m].offset = 33;
    for (int i = 0; i < 128; i++) p->qmap[m].values[i] = i;
    p->qprev[m][0] = p->qprev[m][1] = 500;
  }
  p->bucket_set_size = 0;
}

// ------------------------------------------------------------------------------------------------
struct DBuf {  // grow-only device buffer
  void *p = nullptr;
  size_t cap = 0;
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

enum { ST_INGEST = 0, ST_QUALITY, ST_TOKENIZE, ST_ORDER, ST_EMIT, ST_ENTROPY, ST_COUNT };

// Device buffers of the FRONT stages (ingest .. emit): rows, tokens, tie-break events, sort scratch.  Nothing behind the
// emit stage reads them -- the coder works on the reordered stream and writes the coded one -- so batches whose front stages
// run one after the other on one stream can share a single set (scalce_workspace): with six shards in flight that is the
// difference between 35 GB and 15 GB of HBM per shard (50 M reads x 100 bp).
struct scalce_workspace {
  scalce_ctx *ctx = nullptr;
  u64 row_cap = 0;           // rows the run-wide arrays hold
  u64 piece_rows_cap = 0;    // records one piece may bring (size of the line index)
  DBuf line_end[2], tile[2], packed[2], q[2], namelen, namecell, outlen, names_in, name_in_off, prior_buf;
  DBuf tok_bucket, tok_pos, tie_index, tie_read, tie_off, tie_ncand, cand_bucket, cand_pos, choice;
  DBuf ev_off, ev_bucket, ev_init, ev_sorted, ev_tmp, ev_place, chosen, G, seg, dirty, cand_place, Gseg, cand_fixed;
  DBuf bucket, endv, tokens, counts, bucket_first, bucket_off, chunk, chunk_start;
  DBuf perm_a, perm_b, key_a, key_b, hist, scan_ws, S, run_head, run_hcount, run_rank, runid, run_items_a, run_items_b, run_pos;
  DBuf name_off;
  DBuf tw_cells, tw_cand, tw_bits, tw_base;  // the tie-break in windows (tokenize_windows)
  DBuf tile_mm[2];                           // per text tile: smallest / largest q' symbol (ingest_tiles2_k)
  const void *tile_mm_owner[2] = {nullptr, nullptr};  // the batch whose piece they describe (batches share a workspace)
  const void *walk_owner = nullptr;          // the batch whose first walk tok_bucket / tok_pos hold (scalce_batch_chunk_plan)
  DBuf cell_sorted;                          // name cells in output order (emit stage)
  DBuf qs_shared[2];                         // reordered q' stream of batches that only pass it on (scalce_batch_set_stream_scratch)
  DBuf alt_packed[2], alt_q[2], alt_namelen, alt_namecell, alt_name_in_off, alt_tok_bucket, alt_tok_pos;  // second set of row arrays (scalce_batch_rewindow)
  void free_all() {
    DBuf *all[] = {&line_end[0], &line_end[1], &tile[0], &tile[1], &packed[0], &packed[1], &q[0], &q[1], &namelen, &namecell, &outlen,
                   &names_in, &name_in_off, &prior_buf, &tok_bucket, &tok_pos, &tie_index, &tie_read, &tie_off, &tie_ncand, &cand_bucket,
                   &cand_pos, &choice, &ev_off, &ev_bucket, &ev_init, &ev_sorted, &ev_tmp, &ev_place, &chosen, &G, &seg, &dirty,
                   &cand_place, &Gseg, &cand_fixed, &bucket, &endv, &tokens, &counts, &bucket_first, &bucket_off, &chunk, &chunk_start, &perm_a,
                   &perm_b, &key_a, &key_b, &hist, &scan_ws, &S, &run_head, &run_hcount, &run_rank, &runid, &run_items_a, &run_items_b,
                   &run_pos, &name_off, &tw_cells, &tw_cand, &tw_bits, &tw_base, &tile_mm[0], &tile_mm[1], &cell_sorted, &qs_shared[0], &qs_shared[1],
                   &alt_packed[0], &alt_packed[1], &alt_q[0], &alt_q[1], &alt_namelen, &alt_namecell, &alt_name_in_off, &alt_tok_bucket, &alt_tok_pos};
    for (DBuf *d : all)
      if (d->p) { hipFree(d->p); d->p = nullptr; d->cap = 0; }
    row_cap = piece_rows_cap = 0;
  }
};

struct scalce_batch {
  scalce_workspace *ws;   // front-stage buffers: the batch's own, or shared with other batches (scalce_batch_create_shared)
  bool owns_ws;
  explicit scalce_batch(scalce_workspace *w, bool owns)
      : ws(w), owns_ws(owns), row_cap(w->row_cap), piece_rows_cap(w->piece_rows_cap), line_end(w->line_end), tile(w->tile),
        packed(w->packed), q(w->q), namelen(w->namelen), namecell(w->namecell), outlen(w->outlen), names_in(w->names_in),
        name_in_off(w->name_in_off), prior_buf(w->prior_buf), tok_bucket(w->tok_bucket), tok_pos(w->tok_pos), tie_index(w->tie_index),
        tie_read(w->tie_read), tie_off(w->tie_off), tie_ncand(w->tie_ncand), cand_bucket(w->cand_bucket), cand_pos(w->cand_pos),
        choice(w->choice), ev_off(w->ev_off), ev_bucket(w->ev_bucket), ev_init(w->ev_init), ev_sorted(w->ev_sorted), ev_tmp(w->ev_tmp),
        ev_place(w->ev_place), chosen(w->chosen), G(w->G), seg(w->seg), dirty(w->dirty), cand_place(w->cand_place), Gseg(w->Gseg), cand_fixed(w->cand_fixed),
        bucket(w->bucket), endv(w->endv), tokens(w->tokens), counts(w->counts), bucket_first(w->bucket_first), bucket_off(w->bucket_off),
        chunk(w->chunk), chunk_start(w->chunk_start), perm_a(w->perm_a), perm_b(w->perm_b), key_a(w->key_a), key_b(w->key_b), hist(w->hist),
        scan_ws(w->scan_ws), S(w->S), run_head(w->run_head), run_hcount(w->run_hcount), run_rank(w->run_rank), runid(w->runid),
        run_items_a(w->run_items_a), run_items_b(w->run_items_b), run_pos(w->run_pos), name_off(w->name_off),
        tw_cells(w->tw_cells), tw_cand(w->tw_cand), tw_bits(w->tw_bits), tw_base(w->tw_base), tile_mm(w->tile_mm), cell_sorted(w->cell_sorted) {}
  scalce_ctx *ctx = nullptr;
  scalce_params p;
  u64 max_reads = 0, max_text = 0;
  int nm = 1;
  int L[2] = {0, 0}, stride[2] = {0, 0}, szr[2] = {0, 0}, sz_meta = 1;
  // Rows.  A batch takes its input in one piece (scalce_batch_ingest) or in several (scalce_batch_append): rows
  // [base, base + NP) are the piece being ingested / tokenized, N = base + NP is everything the batch holds.  Packed
  // bases, q', names and tokens are run-wide arrays indexed by row; the text of a piece is dead once it is ingested.
  u64 N = 0, base = 0, NP = 0;
  u64 tok_done = 0, tok_base = 0, tok_n = 0;  // rows tokenized so far / the rows of the tokenization in progress
  u64 &row_cap;              // (of the workspace) rows the run-wide arrays hold
  u64 &piece_rows_cap;       // (of the workspace) records one piece may bring
  bool appending = false;    // the pieces came through scalce_batch_append
  bool lean = false;         // release what a stage no longer needs (runs sized for most of HBM)
  u64 tri_expected[2] = {0, 0};  // trigrams counted so far (tri_check_k)
  u64 names_in_used = 0;     // bytes of the long-name store in use
  u64 S_rows = ~0ull;        // rows the record-size prefix sums in S cover (scalce_batch_chunk_plan), ~0 = stale
  u64 walk_rows = 0;         // rows [0, walk_rows) whose first tokenizer walk (tok_bucket / tok_pos) scalce_batch_chunk_plan has
                             // already done: scalce_batch_tokenize_begin over exactly these rows does not walk them again
  u64 text_bytes[2] = {0, 0};
  const u8 *piece_text[2] = {nullptr, nullptr};  // the piece ingested last (its line index is built on demand)
  bool line_index_ok[2] = {false, false};
  u64 piece_consumed[2] = {0, 0};               // text offset behind the last record taken from it
  bool ingested[2] = {false, false};
  // device state
  DevErr *d_err = nullptr;
  u32 *d_small = nullptr;    // scratch counters: [0..15]
  u64 *d_small64 = nullptr;
  u8 *d_qlut[2] = {nullptr, nullptr};
  int q_affine[2] = {-1, -1};  // the quality map is q - offset for every character: no table lookups in the ingest kernel
  // front-stage buffers (of the workspace)
  DBuf (&line_end)[2], (&tile)[2], (&packed)[2], (&q)[2], &namelen, &namecell, &outlen;
  DBuf &names_in, &name_in_off, &prior_buf;  // names longer than a cell, input order
  DBuf &tok_bucket, &tok_pos, &tie_index, &tie_read, &tie_off, &tie_ncand, &cand_bucket, &cand_pos, &choice;
  DBuf &ev_off, &ev_bucket, &ev_init, &ev_sorted, &ev_tmp, &ev_place, &chosen, &G, &seg, &dirty, &cand_place, &Gseg, &cand_fixed;
  DBuf &bucket, &endv, &tokens, &counts, &bucket_first, &bucket_off, &chunk, &chunk_start;
  // what the coder and the caller read behind the emit stage: the batch's own
  DBuf freq4[2], table[2], qs_own[2], counts_total, bucket_name_bytes, ac_scan;
  // The reordered q' stream: the batch's own (the coder reads it long after the emit stage), or -- scalce_batch_set_stream_scratch,
  // sharded runs: the stream is handed to other ranks right behind the emit stage and the coder reads what came back -- the
  // workspace's, valid until the next batch of the workspace runs its emit stage.
  bool qs_in_ws = false;
  DBuf &qs(int m) { return qs_in_ws ? ws->qs_shared[m] : qs_own[m]; }
  const DBuf &qs(int m) const { return qs_in_ws ? ws->qs_shared[m] : qs_own[m]; }
  const u64 *sorted_keys = nullptr;  // phase-1 keys in output order (order stage), consumed by the emit stage
  u32 key_end_bits = 0, key_bucket_shift = 0, key_bucket_mask = 0;
  DBuf &perm_a, &perm_b, &key_a, &key_b, &hist, &scan_ws, &S, &run_head, &run_hcount, &run_rank, &runid, &run_items_a, &run_items_b, &run_pos;
  DBuf &name_off;
  DBuf &tw_cells, &tw_cand, &tw_bits, &tw_base;
  DBuf (&tile_mm)[2];
  DBuf &cell_sorted;
  bool mm_valid[2] = {false, false};  // tile_mm[m] holds the symbol ranges of the piece ingested last
  bool names_from_sorted_cells = false;
  u32 order_run_members = 0;
  DBuf out_reads[2], out_names, ac_tab[2], ac_tab8[2], ac_cum[2], ac_blocks[2], ac_sizes[2], ac_off[2], ac_desc, out_qual[2];
  AcBlockDesc *ac_desc_host = nullptr;  // block descriptors of the last coder launch this shard led: pinned, so that the
  u32 ac_desc_cap = 0;                  // asynchronous upload never reads memory the next launch is already rewriting
  u32 *perm = nullptr;  // final permutation (points into perm_a or perm_b)
  // host-side results
  u64 out_reads_bytes[2] = {0, 0}, out_names_bytes = 0, out_qual_bytes[2] = {0, 0};
  u32 ntie = 0, nev = 0, ntev = 0, ncand_cap = 0, jacobi_iters = 0, nchunks = 1, sweep_no = 0;
  bool tie_fallback = false;  // the last tie-break ended in tie_sequential_k
  bool tok_open = false;
  int dirty_cur = 0;
  std::vector<uint64_t> explicit_chunks;  // spill-chunk starts given by the caller (sharded runs), else -B rule
  // stage timing
  bool timing = false;
  float stage_ms[ST_COUNT] = {0};
  int stage_launches[ST_COUNT] = {0};
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_group = nullptr;
  // HIP-event pairs around every ac_encode_k launch (the dominant kernel); read by scalce_batch_kernel_ms
  bool ktiming = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> kev;
  size_t kev_used = 0;
  u64 k_in_bytes = 0, k_out_bytes = 0;
  // entropy launched but its result size not read back yet (scalce_batch_entropy_begin / _end): blocks per mate
  u64 *prof_ptr = nullptr;  // SCALCE_AC_PROF of the last rows-coder launch this shard led
  u32 prof_n = 0;
  bool prof_lanes = false;
  u32 ent_pending[2] = {0, 0};
  u32 frame_deferred[2] = {0, 0};  // blocks coded by a grouped launch and not framed yet (entropy_collect frames them)
  // Framing on demand (scalce_batch_set_frame_on_demand): the coded blocks stay where the coder wrote them; entropy_collect
  // only lays the frames out (ac_off: where block k's [u32 size][bytes] begins in the stream).  The stream itself is
  // produced on its way out -- scalce_batch_qual_window, into device or pinned host memory -- or, for callers that ask for
  // SCALCE_OUT_QUAL as a device pointer, once, at that moment.
  // Bytes per block of the coder's output buffers.  The reference gives every block 10 MiB (arithmetic.cpp:301); sized like that
  // a 50 M-read shard holds 5 GB of which 2.9 are used.  ac_prepare sizes the stride from what the table says coding its own
  // counts costs (+ 8 % + 64 KiB); a block that outgrows it reports E_ACOVERFLOW and the shard is coded again at the full
  // stride when it is collected (entropy_recode_full) -- same bytes, one launch later.
  // One row per read (single-end runs, read lengths the tile ingest takes): q[0] holds rows of `qstride[0]` bytes -- q' | a copy
  // of the packed words: 128 bytes = one aligned line for a 100 bp read -- and the emit stage gathers a record's q' and bases
  // with ONE random line (emit_reads_k<true>).  Otherwise qstride[m] = L[m]: rows back to back.
  bool fused = false;
  u32 qstride[2] = {0, 0}, row_cell_off = 0, row_pwords = 0;
  DBuf q_compact, fuse_q, fuse_cells;  // SCALCE_OUT_QINPUT of fused rows on request; classic arrays of a piece the indexed kernels took
  u64 ac_stride[2] = {0, 0};
  // Coding in place (scalce_batch_set_code_in_place): the coder's output goes over the symbols it has consumed -- block k's bytes
  // begin where block k's symbols began, ac_base = the reordered stream itself, ac_stride = 10 MiB -- and the batch holds no
  // block buffers at all (3.2 GB per 50 M reads of 100 bp).  A block whose output would catch up with its input reports
  // E_ACOVERFLOW; its symbols are gone by then, so the shard is run again FROM ITS TEXT with buffers of its own
  // (entropy_rerun_from_text: the caller keeps the text of a shard in place until the shard is collected).
  bool code_in_place = false, in_place_suspended = false;
  u64 reruns = 0;                          // shards run again from their text (entropy_rerun_from_text)
  bool in_place_now[2] = {false, false};   // the last launch coded this mate's stream in place
  u8 *ac_base[2] = {nullptr, nullptr};     // block k of the last launch: ac_base + k * ac_stride
  DBuf ac_log[2];                          // carry notes of ac_encode_lanes_k when the block's buffer has no room for them
  const u8 *ac_last_sym[2] = {nullptr, nullptr};  // what the last launch coded (for the recode)
  u64 ac_last_nsym[2] = {0, 0};
  bool frame_on_demand = false;
  u32 frame_virtual[2] = {0, 0};   // blocks whose frames are laid out but not copied (0: out_qual holds the stream)
  std::vector<u64> frame_off_host[2];  // where block k's frame begins (host copy, taken when the stage is collected)
  // symbol stream to code per mate: the shard's own reordered stream, or one the caller assembled (sharded runs)
  const u8 *ent_sym[2] = {nullptr, nullptr};
  u64 ent_nsym[2] = {0, 0};
  bool ent_external[2] = {false, false};
};

static int ensure(scalce_batch *b, DBuf &d, size_t bytes) {
  if (bytes <= d.cap) return SCALCE_OK;
  // (an allocation synchronises the whole device: in a pipeline it waits for every coder that is running.  SCALCE_DEBUG_ALLOC=1
  //  names the ones that still happen after the warm-up)
  static const bool dbg = getenv("SCALCE_DEBUG_ALLOC") != nullptr;
  if (dbg) fprintf(stderr, "scalce: batch %p grows a buffer from %zu to %zu bytes\n", (void *)b, d.cap, bytes);
  if (d.p) hipFree(d.p);
  d.p = nullptr; d.cap = 0;
  bytes = (bytes + 255) & ~size_t(255);
  hipError_t e = hipMalloc(&d.p, bytes);
  if (e != hipSuccess) {
    set_err(b->ctx, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return SCALCE_ERR_HIP;
  }
  d.cap = bytes;
  return SCALCE_OK;
}
#define ENSURE(b, buf, bytes) do { int rc_ = ensure(b, buf, bytes); if (rc_) return rc_; } while (0)
// the same for run-wide arrays that grow while a run is ingested piece by piece: the first `used` bytes survive
static int ensure_keep(scalce_batch *b, DBuf &d, size_t bytes, size_t used, hipStream_t s) {
  if (bytes <= d.cap) return SCALCE_OK;
  if (!d.p || !used) return ensure(b, d, bytes);
  size_t want = d.cap + d.cap / 2;
  if (want < bytes) want = bytes;
  want = (want + 255) & ~size_t(255);
  void *np = nullptr;
  hipError_t e = hipMalloc(&np, want);
  if (e != hipSuccess && want > bytes) { want = (bytes + 255) & ~size_t(255); e = hipMalloc(&np, want); }
  if (e != hipSuccess) {
    set_err(b->ctx, "hipMalloc(%zu) failed while growing a run-wide array: %s", want, hipGetErrorString(e));
    return SCALCE_ERR_HIP;
  }
  if ((e = hipMemcpyAsync(np, d.p, used, hipMemcpyDeviceToDevice, s)) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess) {
    hipFree(np);
    set_err(b->ctx, "growing a run-wide array: %s", hipGetErrorString(e));
    return SCALCE_ERR_HIP;
  }
  hipFree(d.p);
  d.p = np;
  d.cap = want;
  return SCALCE_OK;
}
static void release(DBuf &d) {
  if (d.p) hipFree(d.p);
  d.p = nullptr;
  d.cap = 0;
}

static void free_all(scalce_batch *b) {
  DBuf *all[] = {&b->freq4[0], &b->freq4[1], &b->table[0], &b->table[1], &b->qs_own[0], &b->qs_own[1], &b->counts_total, &b->bucket_name_bytes,
                 &b->ac_scan, &b->out_reads[0], &b->out_reads[1], &b->out_names, &b->ac_tab[0], &b->ac_cum[0], &b->ac_blocks[0],
                 &b->ac_sizes[0], &b->ac_off[0], &b->ac_tab[1], &b->ac_cum[1], &b->ac_blocks[1], &b->ac_sizes[1], &b->ac_off[1],
                 &b->ac_desc, &b->out_qual[0], &b->out_qual[1], &b->ac_tab8[0], &b->ac_tab8[1], &b->q_compact, &b->fuse_q, &b->fuse_cells,
                 &b->ac_log[0], &b->ac_log[1]};
  for (DBuf *d : all)
    if (d->p) { hipFree(d->p); d->p = nullptr; d->cap = 0; }
  if (b->owns_ws) { b->ws->free_all(); delete b->ws; }
  if (b->d_err) hipFree(b->d_err);
  if (b->d_small) hipFree(b->d_small);
  if (b->d_small64) hipFree(b->d_small64);
  for (int m = 0; m < 2; m++) if (b->d_qlut[m]) hipFree(b->d_qlut[m]);
  if (b->ac_desc_host) hipHostFree(b->ac_desc_host);
  if (b->ev0) hipEventDestroy(b->ev0);
  if (b->ev1) hipEventDestroy(b->ev1);
  if (b->ev_group) hipEventDestroy(b->ev_group);
  for (auto &pr : b->kev) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
}

static inline int sz_read(int l) { return (l + 3) / 4; }
// room behind the reordered stream: the last block of a stream coded in place may write this much more than it holds symbols
constexpr size_t AC_INPLACE_PAD = 65536;

// run-wide arrays indexed by row: room for `rows` of them, the first `used` rows kept
static int reserve_rows(scalce_batch *b, u64 rows, u64 used, hipStream_t s) {
  if (rows <= b->row_cap) return SCALCE_OK;
  if (rows >= (1ull << 32) - 64) { set_err(b->ctx, "a batch holds fewer than 2^32 reads"); return SCALCE_ERR_CAPACITY; }
  if (b->row_cap && rows < b->row_cap + b->row_cap / 4) rows = b->row_cap + b->row_cap / 4;  // grow in steps
  int rc;
  for (int m = 0; m < b->nm; m++) {
    if ((rc = ensure_keep(b, b->packed[m], (size_t)b->stride[m] * rows + 64, (size_t)b->stride[m] * used, s))) return rc;
    if ((rc = ensure_keep(b, b->q[m], (size_t)b->qstride[m] * rows + 64, (size_t)b->qstride[m] * used, s))) return rc;
  }
  if ((rc = ensure_keep(b, b->namelen, rows + 64, used, s))) return rc;
  if (b->p.use_names && (rc = ensure_keep(b, b->namecell, 16 * (rows + 8), 16 * used, s))) return rc;
  if (b->name_in_off.p && (rc = ensure_keep(b, b->name_in_off, sizeof(u64) * (rows + 2), sizeof(u64) * used, s))) return rc;
  if ((rc = ensure_keep(b, b->bucket, sizeof(u32) * (rows + 1), sizeof(u32) * used, s))) return rc;
  if ((rc = ensure_keep(b, b->endv, sizeof(u16) * (rows + 1), sizeof(u16) * used, s))) return rc;
  if ((rc = ensure_keep(b, b->tokens, sizeof(int32_t) * 2 * (rows + 1), sizeof(int32_t) * 2 * used, s))) return rc;
  b->row_cap = rows;
  return SCALCE_OK;
}

extern "C" int scalce_workspace_create(scalce_ctx *c, scalce_workspace **out) {
  if (!c || !out) return SCALCE_ERR_ARG;
  *out = new scalce_workspace();
  (*out)->ctx = c;
  return SCALCE_OK;
}
extern "C" void scalce_workspace_destroy(scalce_workspace *w) {
  if (!w) return;
  hipSetDevice(w->ctx->device);
  w->free_all();
  delete w;
}

static int batch_create(scalce_ctx *c, const scalce_params *p, uint64_t max_reads, uint64_t max_text, scalce_workspace *shared,
                        scalce_batch **out);
extern "C" int scalce_batch_create(scalce_ctx *c, const scalce_params *p, uint64_t max_reads, uint64_t max_text,
                                   scalce_batch **out) {
  return batch_create(c, p, max_reads, max_text, nullptr, out);
}
extern "C" int scalce_batch_create_shared(scalce_ctx *c, const scalce_params *p, uint64_t max_reads, uint64_t max_text,
                                          scalce_workspace *w, scalce_batch **out) {
  if (!w || w->ctx != c) return SCALCE_ERR_ARG;
  return batch_create(c, p, max_reads, max_text, w, out);
}
static int batch_create(scalce_ctx *c, const scalce_params *p, uint64_t max_reads, uint64_t max_text, scalce_workspace *shared,
                        scalce_batch **out) {
  if (!c || !p || !out) return SCALCE_ERR_ARG;
  if (!c->have_patterns) { set_err(c, "load a core table first"); return SCALCE_ERR_ARG; }
  if (p->read_len[0] <= 0 || p->read_len[0] > 2498 || (p->paired && (p->read_len[1] <= 0 || p->read_len[1] > 2498))) {
    set_err(c, "read_len must be set (1..2498: the reference reads lines into MAXLINE = 2500 bytes, const.h:87)");
    return SCALCE_ERR_ARG;
  }
  if (max_reads >= (1ull << 32) - 64) { set_err(c, "a shard holds fewer than 2^32 reads"); return SCALCE_ERR_CAPACITY; }
  HIP_TRY(c, hipSetDevice(c->device));
  scalce_workspace *w = shared;
  if (!w) { w = new scalce_workspace(); w->ctx = c; }
  scalce_batch *b = new scalce_batch(w, shared == nullptr);
  b->ctx = c;
  b->p = *p;
  b->max_reads = max_reads;
  b->max_text = max_text;
  b->nm = p->paired ? 2 : 1;
  for (int m = 0; m < b->nm; m++) {
    b->L[m] = p->read_len[m];
    b->szr[m] = sz_read(b->L[m]);
    b->stride[m] = ((b->szr[m] + 1 + 15) / 16) * 16;  // one spare zero byte for 16-bit digit windows
  }
  b->sz_meta = b->L[0] > 255 ? 2 : 1;  // reads.cpp:106-108
  b->qstride[0] = (u32)b->L[0];
  b->qstride[1] = (u32)b->L[1];
  if (b->nm == 1 && (b->L[0] & 3) == 0 && b->L[0] >= 16 && b->L[0] <= 160) {
    b->fused = true;
    b->row_cell_off = (u32)b->L[0];   // where the packed words begin
    b->row_pwords = (u32)(b->L[0] + 15) / 16;
    b->qstride[0] = (b->row_cell_off + 4 * b->row_pwords + 15) / 16 * 16;
  }
  *out = b;
  HIP_TRY(c, hipMalloc(&b->d_err, sizeof(DevErr)));
  HIP_TRY(c, hipMemset(b->d_err, 0, sizeof(DevErr)));
  HIP_TRY(c, hipMalloc(&b->d_small, 64 * sizeof(u32)));
  HIP_TRY(c, hipMalloc(&b->d_small64, 512 * sizeof(u64)));
  HIP_TRY(c, hipEventCreate(&b->ev0));
  HIP_TRY(c, hipEventCreate(&b->ev1));
  for (int m = 0; m < b->nm; m++) {
    u8 lut[128];
    bool identity = p->qmap[m].offset >= 0 && p->qmap[m].offset < 128;
    for (int i = 0; i < 128; i++) {
      lut[i] = (u8)((p->qmap[m].values[i] - p->qmap[m].offset) & 255);
      identity = identity && p->qmap[m].values[i] == i;
    }
    b->q_affine[m] = identity ? (int)p->qmap[m].offset : -1;
    HIP_TRY(c, hipMalloc(&b->d_qlut[m], 128));
    HIP_TRY(c, hipMemcpy(b->d_qlut[m], lut, 128, hipMemcpyHostToDevice));
    ENSURE(b, b->freq4[m], sizeof(u64) * 512000);
    ENSURE(b, b->table[m], sizeof(u32) * 512000);
  }
  // a record is at least "@x", L bases, "+", L qualities and four newlines: what one piece of max_text bytes can bring
  const u64 per_piece = max_text / (2 * (u64)b->L[0] + 7) + 2;
  b->piece_rows_cap = per_piece < max_reads ? per_piece : max_reads;
  // (the line index of a piece, 32 bytes per record, is only built when something asks for it: ensure_line_index)
  { int rc = reserve_rows(b, max_reads, 0, nullptr); if (rc) return rc; }
  ENSURE(b, b->scan_ws, sizeof(u64) * (scan_ws_elems(4 * b->piece_rows_cap + 1024) + 4096));
  return SCALCE_OK;
}

extern "C" void scalce_batch_destroy(scalce_batch *b) {
  if (!b) return;
  hipSetDevice(b->ctx->device);
  free_all(b);
  delete b;
}

struct StageTimer {
  scalce_batch *b;
  int st;
  hipStream_t s;
  StageTimer(scalce_batch *b_, int st_, hipStream_t s_) : b(b_), st(st_), s(s_) {
    if (b->timing) hipEventRecord(b->ev0, s);
  }
  ~StageTimer() {
    if (b->timing) {
      hipEventRecord(b->ev1, s);
      hipEventSynchronize(b->ev1);
      float ms = 0;
      hipEventElapsedTime(&ms, b->ev0, b->ev1);
      b->stage_ms[st] += ms;
      b->stage_launches[st]++;
    }
  }
};

static int launch_failed(scalce_ctx *c);
static int read_u32(scalce_batch *b, const u32 *d, u32 *h, int n, hipStream_t s) {
  { int rc = launch_failed(b->ctx); if (rc) return rc; }
  HIP_TRY(b->ctx, hipMemcpyAsync(h, d, sizeof(u32) * n, hipMemcpyDeviceToHost, s));
  HIP_TRY(b->ctx, hipStreamSynchronize(s));
  return SCALCE_OK;
}
static int read_u64(scalce_batch *b, const u64 *d, u64 *h, int n, hipStream_t s) {
  { int rc = launch_failed(b->ctx); if (rc) return rc; }
  HIP_TRY(b->ctx, hipMemcpyAsync(h, d, sizeof(u64) * n, hipMemcpyDeviceToHost, s));
  HIP_TRY(b->ctx, hipStreamSynchronize(s));
  return SCALCE_OK;
}

static int launch_failed(scalce_ctx *c);
static int check_device_error(scalce_batch *b, hipStream_t s) {
  { int rc = launch_failed(b->ctx); if (rc) return rc; }
  DevErr e;
  HIP_TRY(b->ctx, hipMemcpyAsync(&e, b->d_err, sizeof e, hipMemcpyDeviceToHost, s));
  HIP_TRY(b->ctx, hipStreamSynchronize(s));
  if (e.code == E_NONE) return SCALCE_OK;
  static const char *names[] = {"", "line count is not a multiple of 4 or the text does not end in a newline",
                                "read or quality line length differs from read_length (compress.cpp:628-634)",
                                "read name longer than 255 bytes or empty name line",
                                "quality symbol >= 80 after mapping (arithmetic.h:47)",
                                "arithmetic-coded block larger than the reference's 10 MiB buffer (arithmetic.cpp:101)",
                                "mates have different record counts", "internal"};
  set_err(b->ctx, "(ERROR) %s [record/block %llu, aux %u]", names[e.code < 8 ? e.code : 7],
          (unsigned long long)e.where, e.aux);
  hipMemsetAsync(b->d_err, 0, sizeof(DevErr), s);
  return SCALCE_ERR_FORMAT;
}

// A launch that the runtime refuses (a grid beyond 2^32 threads, say) must not pass for a kernel that ran: HIP's "last
// error" is overwritten by the next call that succeeds, so it is looked at right behind every launch and kept until
// scalce_batch_finish / the next read-back reports it.
static thread_local hipError_t g_launch_err = hipSuccess;
static thread_local const char *g_launch_what = "";
#define LAUNCH(kernel, grid, block, shmem, stream, ...)                                              \
  do {                                                                                               \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, stream, __VA_ARGS__);                 \
    const hipError_t le_ = hipGetLastError();                                                        \
    if (le_ != hipSuccess && g_launch_err == hipSuccess) { g_launch_err = le_; g_launch_what = #kernel; } \
  } while (0)
static int launch_failed(scalce_ctx *c) {
  if (g_launch_err == hipSuccess) return SCALCE_OK;
  set_err(c, "launch of %s failed: %s", g_launch_what, hipGetErrorString(g_launch_err));
  g_launch_err = hipSuccess;
  return SCALCE_ERR_HIP;
}
static inline u32 cdiv(u64 a, u64 b) {
  const u64 q = (a + b - 1) / b;
  return q > 0x7FFFFFFFull ? 0x7FFFFFFFu : (u32)q;  // callers whose grids can get there use grid-stride kernels
}

// ---- stage 0: ingest ------------------------------------------------------------------------------
// newline count of one mate's text; the per-tile bases stay in b->tile[mate] for piece_unpack
static int piece_count(scalce_batch *b, int mate, const u8 *d_text, u64 nbytes, hipStream_t s, u64 *nlines, u8 *last) {
  scalce_ctx *c = b->ctx;
  if (((uintptr_t)d_text & 15) != 0) { set_err(c, "FASTQ text must be 16-byte aligned"); return SCALCE_ERR_ARG; }
  if (nbytes > b->max_text) b->max_text = nbytes;  // max_text only sizes the first allocations; everything below grows
  b->text_bytes[mate] = nbytes;
  const u32 ntiles = cdiv(nbytes, IDX_TILE);
  ENSURE(b, b->tile[mate], (ntiles + 8) * sizeof(u64));
  ENSURE(b, b->scan_ws, (scan_ws_elems(ntiles) + 64) * sizeof(u64));
  u64 *tile = b->tile[mate].as<u64>();  // [ntiles] counts -> bases
  *nlines = 0;
  *last = '\n';
  if (ntiles) {
    LAUNCH(index_count_k, ntiles, IDX_THREADS, 0, s, d_text, nbytes, tile);
    exclusive_scan<u64>(LoadAs<u64, u64>{tile}, ntiles, StoreTo<u64>{tile}, b->scan_ws.as<u64>(), b->d_small64 + 4 + mate, s);
    HIP_TRY(c, hipMemcpyAsync(nlines, b->d_small64 + 4 + mate, sizeof(u64), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(last, d_text + nbytes - 1, 1, hipMemcpyDeviceToHost, s));
  }
  return SCALCE_OK;
}

// the line index of the piece (index_write_k), built when something needs it: the indexed unpack kernels, names longer
// than a cell, scalce_batch_text_offset
static int ensure_line_index(scalce_batch *b, int mate, hipStream_t s) {
  if (b->line_index_ok[mate]) return SCALCE_OK;
  const u64 nrec = b->NP, nbytes = b->text_bytes[mate];
  if (nrec > b->piece_rows_cap || !b->line_end[mate].p) {
    if (nrec > b->piece_rows_cap) b->piece_rows_cap = nrec;
    ENSURE(b, b->line_end[mate], sizeof(u64) * 4 * (b->piece_rows_cap + 1));
  }
  const u32 ntiles = cdiv(nbytes, IDX_TILE);
  if (ntiles) LAUNCH(index_write_k, ntiles, IDX_THREADS, 0, s, b->piece_text[mate], nbytes, b->tile[mate].as<u64>(), b->line_end[mate].as<u64>(), 4 * nrec);
  b->line_index_ok[mate] = true;
  return SCALCE_OK;
}

// the first `nrec` records of the text -> rows [base, base + nrec): 2-bit bases, q', names
// (A one-pass variant -- no count pass, the tiles' line bases by decoupled look-back between the workgroups -- was byte-exact
// and slower: 9.2 ms against 1.7 + 6.2 at 50 M x 100 bp, rounds 3-4; removed in round 5.)
static int piece_unpack(scalce_batch *b, int mate, const u8 *d_text, u64 nbytes, u64 nrec, hipStream_t s) {
  scalce_ctx *c = b->ctx;
  b->piece_text[mate] = d_text;
  b->line_index_ok[mate] = false;
  b->piece_consumed[mate] = 0;
  if (!nrec) return SCALCE_OK;
  UnpackArgs a;
  a.text = d_text; a.nbytes = nbytes; a.line_end = nullptr; a.nrec = nrec;
  a.L = b->L[mate]; a.stride = b->stride[mate]; a.mate = mate; a.use_names = b->p.use_names; a.no_ac = b->p.no_ac;
  a.packed = b->packed[mate].as<u8>() + b->base * (u64)b->stride[mate];
  a.q = b->q[mate].as<u8>() + b->base * (u64)b->qstride[mate];
  a.qstride = b->qstride[mate];
  a.cellstride = 16;
  a.packed2 = nullptr;
  a.namelen = b->namelen.as<u8>() + b->base;
  a.namecell = (mate == 0 && b->p.use_names) ? b->namecell.as<u8>() + 16 * b->base : nullptr;
  const bool fused_rows = b->fused && mate == 0;
  if (fused_rows) a.packed2 = a.q + b->row_cell_off;  // a copy of the packed words lies behind the row's q'
  a.qlut = b->d_qlut[mate]; a.err = b->d_err;
  a.q_affine = b->q_affine[mate];
  a.max_namelen = b->d_small + 16;
  u32 *slow = b->d_small + 17;
  u64 *d_consumed = b->d_small64 + 2 + mate;  // (slots 1..3 are the emit stage's, long after this)
  if (mate == 0) HIP_TRY(c, hipMemsetAsync(b->d_small + 16, 0, 2 * sizeof(u32), s));
  else HIP_TRY(c, hipMemsetAsync(slow, 0, sizeof(u32), s));
  // one pass behind the count for the usual read lengths (ingest_tiles_k); the indexed kernels otherwise, and when a
  // record turns out not to fit the tile overlap
  bool fused = a.L >= 16 && a.L <= 160 && !getenv("SCALCE_INGEST_INDEXED");
  u32 flags[2] = {0, 0};
  b->mm_valid[mate] = false;
  if (fused) {
    IngestArgs ia;
    ia.u = a;
    ia.tile_base = b->tile[mate].as<u64>();
    ia.consumed = d_consumed;
    ia.slow = slow;
    {
      const u32 ntiles2 = cdiv(nbytes, ING_TILE);
      ENSURE(b, b->tile_mm[mate], sizeof(u16) * ((size_t)ntiles2 + 8));
      Ingest2Args ga;
      ga.i = ia;
      const u64 S = (u64)a.stride / 4, W = ((u64)a.L + 15) / 16;
      ga.magic_s = ((1ull << 32) + S - 1) / S;
      ga.magic_w = ((1ull << 32) + W - 1) / W;
      ga.step_ks = (u32)(ING_THREADS / S); ga.step_rs = (u32)(ING_THREADS % S);
      ga.step_kw = (u32)(ING_THREADS / W); ga.step_rw = (u32)(ING_THREADS % W);
      ga.tile_minmax = b->tile_mm[mate].as<u16>();
      ga.ticket = nullptr; ga.status = nullptr; ga.tile_base_out = nullptr;
      LAUNCH(ingest_tiles2_k<false>, ntiles2, ING_THREADS, 0, s, ga);
      b->mm_valid[mate] = true;
      b->ws->tile_mm_owner[mate] = b;
    }
    { int rc = read_u32(b, b->d_small + 16, flags, 2, s); if (rc) return rc; }
    if (flags[1]) { fused = false; b->mm_valid[mate] = false; }  // a record longer than the overlap: redo the piece the indexed way
  }
  if (!fused) {
    { int rc = ensure_line_index(b, mate, s); if (rc) return rc; }
    a.line_end = b->line_end[mate].as<u64>();
    if (mate == 0) HIP_TRY(c, hipMemsetAsync(b->d_small + 16, 0, sizeof(u32), s));
    u8 *row_q = a.q;
    if (fused_rows) {  // the indexed kernels write rows back to back: into an array of the piece's own, fused behind them
      ENSURE(b, b->fuse_q, (size_t)a.L * nrec + 64);
      a.q = b->fuse_q.as<u8>();
      a.qstride = (u32)a.L;
      a.packed2 = nullptr;
    }
    if ((size_t)UNP_RPB * a.L <= (size_t)UNP_Q_CAP)
      LAUNCH(unpack_tiled_k, cdiv(nrec, UNP_RPB), 2 * UNP_RPB, unp_text_cap(a.L) + 32 + unp_q_cap(a.L), s, a);
    else LAUNCH(unpack_k, cdiv(nrec, 256), 256, 0, s, a);
    if (fused_rows)
      LAUNCH(fuse_rows_k, cdiv(nrec, 256), 256, 0, s, nrec, b->fuse_q.as<u8>(), (u32)a.L, (const u8 *)nullptr, a.packed, (u32)a.stride, b->row_pwords,
             row_q, b->qstride[0], b->row_cell_off);
    LAUNCH(last_record_end_k, 1, 1, 0, s, a.line_end, nrec, d_consumed);
    { int rc = read_u32(b, b->d_small + 16, flags, 1, s); if (rc) return rc; }
  }
  { u64 v = 0; int rc = read_u64(b, d_consumed, &v, 1, s); if (rc) return rc; b->piece_consumed[mate] = v; }
  if (mate == 0 && b->p.use_names) {
    // names that do not fit their 16-byte cell go to the long-name store (input order): the text is not needed again
    const u32 maxlen = flags[0];
    if (maxlen > 15) {
      { int rc = ensure_line_index(b, mate, s); if (rc) return rc; }
      if (!b->name_in_off.p) {
        ENSURE(b, b->name_in_off, sizeof(u64) * (b->row_cap + 2));
        HIP_TRY(c, hipMemsetAsync(b->name_in_off.p, 0, sizeof(u64) * (b->row_cap + 2), s));
      }
      ENSURE(b, b->scan_ws, sizeof(u64) * (scan_ws_elems(nrec) + 64));
      u64 *off = b->name_in_off.as<u64>() + b->base;
      exclusive_scan<u64>(LongNameLen{a.namelen}, nrec, StoreTo<u64>{off}, b->scan_ws.as<u64>(), b->d_small64 + 6, s);
      u64 total = 0;
      { int rc = read_u64(b, b->d_small64 + 6, &total, 1, s); if (rc) return rc; }
      { int rc = ensure_keep(b, b->names_in, b->names_in_used + total + 64, b->names_in_used, s); if (rc) return rc; }
      LAUNCH(long_names_k, cdiv(nrec, 256), 256, 0, s, nrec, d_text, b->line_end[mate].as<u64>(), a.namelen, off, b->names_in_used, b->names_in.as<u8>());
      b->names_in_used += total;
    }
  }
  return SCALCE_OK;
}

static void batch_restart(scalce_batch *b) {
  b->N = b->base = b->NP = 0;
  b->S_rows = ~0ull;
  b->walk_rows = 0;
  b->tok_done = b->tok_base = b->tok_n = 0;
  b->appending = false;
  b->names_in_used = 0;
  b->tri_expected[0] = b->tri_expected[1] = 0;
  b->ingested[0] = b->ingested[1] = false;
  // (frames laid out but not copied belong to the shard that is being replaced: scalce_batch_qual_window must not serve them)
  for (int m = 0; m < 2; m++) { b->frame_virtual[m] = 0; b->frame_off_host[m].clear(); }
}

extern "C" int scalce_batch_reset(scalce_batch *b) {
  if (!b) return SCALCE_ERR_ARG;
  batch_restart(b);
  return SCALCE_OK;
}

extern "C" int scalce_batch_ingest(scalce_batch *b, int mate, const uint8_t *d_text, uint64_t nbytes, void *stream) {
  if (!b || mate < 0 || mate >= b->nm || !d_text) return SCALCE_ERR_ARG;
  scalce_ctx *c = b->ctx;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(c, hipSetDevice(c->device));
  StageTimer tm(b, ST_INGEST, s);
  if (mate == 0) batch_restart(b);  // one piece = the whole shard
  u64 nlines = 0;
  u8 last = '\n';
  { int rc = piece_count(b, mate, d_text, nbytes, s, &nlines, &last); if (rc) return rc; }
  HIP_TRY(c, hipStreamSynchronize(s));
  if ((nlines & 3) || last != '\n') {
    set_err(c, "(ERROR) FASTQ text has %llu lines (not a multiple of 4) or no trailing newline", (unsigned long long)nlines);
    return SCALCE_ERR_FORMAT;
  }
  const u64 nrec = nlines / 4;
  if (nrec > b->max_reads) { set_err(c, "%llu records exceed the batch capacity", (unsigned long long)nrec); return SCALCE_ERR_CAPACITY; }
  if (mate == 0) { b->N = b->NP = nrec; }
  else if (nrec != b->N) { set_err(c, "(ERROR) mates have different record counts"); return SCALCE_ERR_FORMAT; }
  { int rc = piece_unpack(b, mate, d_text, nbytes, nrec, s); if (rc) return rc; }
  b->ingested[mate] = true;
  return SCALCE_OK;
}

// Streaming form of the record reader: the next piece of the read stream goes BEHIND the rows the batch already holds.
// As many complete records as both mates' pieces hold are taken (compress.cpp:614-666 reads the mates in step);
// consumed[m] says how many bytes of each piece that was, the caller hands the rest in again in front of the next
// piece.  Ingest, quality counters and the tie-break of the new rows (against all rows before them) run here; order,
// emit and entropy run once, over everything, when the caller has no more input.
static int first_walk(scalce_batch *b, u64 row0, u64 n, u64 tok_row0, hipStream_t s);
static int entropy_rerun_from_text(scalce_batch *b, hipStream_t s);
// the ingest half of scalce_batch_append: as many complete records as both mates' pieces hold -> rows [N, N + nrec)
static int ingest_piece(scalce_batch *b, const uint8_t *const text[2], const u64 nbytes[2], bool final_piece, uint64_t consumed[2], hipStream_t s) {
  scalce_ctx *c = b->ctx;
  consumed[0] = consumed[1] = 0;
  u64 nlines[2] = {0, 0}, nrec = ~0ull;
  u8 last[2] = {'\n', '\n'};
  StageTimer tm(b, ST_INGEST, s);
  for (int m = 0; m < b->nm; m++)
    if (nbytes[m]) { int rc = piece_count(b, m, text[m], nbytes[m], s, &nlines[m], &last[m]); if (rc) return rc; }
  HIP_TRY(c, hipStreamSynchronize(s));
  for (int m = 0; m < b->nm; m++) nrec = nlines[m] / 4 < nrec ? nlines[m] / 4 : nrec;
  if (final_piece) {
    for (int m = 0; m < b->nm; m++)
      if ((nlines[m] & 3) || last[m] != '\n') {
        set_err(c, "(ERROR) FASTQ text has %llu lines (not a multiple of 4) or no trailing newline", (unsigned long long)(4 * b->N + nlines[m]));
        return SCALCE_ERR_FORMAT;
      }
    if (b->nm == 2 && nlines[0] != nlines[1]) { set_err(c, "(ERROR) mates have different record counts"); return SCALCE_ERR_FORMAT; }
  }
  b->base = b->N;
  b->NP = nrec;
  if (b->base + nrec >= (1ull << 32) - 64) { set_err(c, "a batch holds fewer than 2^32 reads"); return SCALCE_ERR_CAPACITY; }
  { int rc = reserve_rows(b, b->base + nrec, b->base, s); if (rc) return rc; }
  for (int m = 0; m < b->nm; m++) {
    int rc = piece_unpack(b, m, text[m], nbytes[m], nrec, s);
    if (rc) return rc;
    b->ingested[m] = true;
    consumed[m] = nrec ? b->piece_consumed[m] : 0;  // behind the newline that ends the last record taken
  }
  HIP_TRY(c, hipStreamSynchronize(s));
  b->N = b->base + nrec;
  return SCALCE_OK;
}

extern "C" int scalce_batch_append(scalce_batch *b, const uint8_t *d_text1, uint64_t n1, const uint8_t *d_text2, uint64_t n2,
                                   int flags, uint64_t consumed[2], void *stream) {
  const int final_piece = flags & SCALCE_APPEND_FINAL;
  if (!b || !consumed || (n1 && !d_text1) || (b->nm == 2 && n2 && !d_text2)) return SCALCE_ERR_ARG;
  scalce_ctx *c = b->ctx;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(c, hipSetDevice(c->device));
  if (b->tok_open) { set_err(c, "a tokenization is still open"); return SCALCE_ERR_ARG; }
  if (!b->appending) { batch_restart(b); b->appending = true; }
  const uint8_t *text[2] = {d_text1, d_text2};
  const u64 nbytes[2] = {n1, b->nm == 2 ? n2 : 0};
  int rc = ingest_piece(b, text, nbytes, final_piece != 0, consumed, s);
  if (rc) return rc;
  if (!(flags & SCALCE_APPEND_NO_QUALITY) && (rc = scalce_batch_quality(b, stream))) return rc;
  if (!(flags & SCALCE_APPEND_NO_TOKENIZE) && (rc = scalce_batch_tokenize(b, nullptr, stream))) return rc;
  return SCALCE_OK;
}

// Sharded runs: the rows this rank holds change at both ends (rank boundaries move to spill-chunk boundaries, sharded.cpp) --
// rows [keep_first, keep_first + keep_rows) stay, the records of `front` go in front of them, those of `back` behind (FASTQ text,
// whole records, either may be empty).  Rounds 1-4 rebuilt the whole range from text: a second ingest and a second first walk of
// every row (+21 ms per 50 M-read shard).  Here the rows that stay stay: the run-wide row arrays and the first walk's tokens are
// swapped against a second set in the workspace, the new set takes [front | kept | back] -- only the moved records are ingested
// and walked, the kept rows are one device copy (rows, name cells, tokens: ~185 bytes per read); nothing at all is copied when
// only the back end moves.  Quality statistics are NOT touched: every record was counted by the rank that ingested it first.
extern "C" int scalce_batch_rewindow(scalce_batch *b, uint64_t keep_first, uint64_t keep_rows, const uint8_t *const front[2],
                                     const uint64_t front_bytes[2], const uint8_t *const back[2], const uint64_t back_bytes[2], void *stream) {
  if (!b || !front || !back || !front_bytes || !back_bytes || keep_first + keep_rows > b->N) return SCALCE_ERR_ARG;
  scalce_ctx *c = b->ctx;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(c, hipSetDevice(c->device));
  if (b->tok_open || b->tok_done) { set_err(c, "rewindow: the rows have been tokenized already"); return SCALCE_ERR_ARG; }
  scalce_workspace *w = b->ws;
  const bool walked = b->walk_rows == b->N && w->walk_owner == b && b->N > 0;  // (scalce_batch_chunk_plan has been here)
  const bool have_front = front_bytes[0] != 0;
  uint64_t used[2];
  const u64 fb[2] = {front_bytes[0], b->nm == 2 ? front_bytes[1] : 0}, bb[2] = {back_bytes[0], b->nm == 2 ? back_bytes[1] : 0};
  u64 nfront = 0;
  b->S_rows = ~0ull;
  b->appending = true;
  // rows the new range may hold (a record is at least "@x", L bases, "+", L qualities and four newlines): the tokens of the
  // first walk are sized for it BEFORE the kept ones move -- growing the array later would lose them
  const u64 rows_bound = keep_rows + fb[0] / (2 * (u64)b->L[0] + 7) + bb[0] / (2 * (u64)b->L[0] + 7) + 8;
  if (!have_front && keep_first == 0) {
    b->N = keep_rows;  // only the back end moves: rows beyond keep_rows are dropped where they lie
    if (walked) {
      int rc = ensure_keep(b, b->tok_bucket, sizeof(u32) * (rows_bound + 1), sizeof(u32) * keep_rows, s);
      if (!rc) rc = ensure_keep(b, b->tok_pos, sizeof(u32) * (rows_bound + 1), sizeof(u32) * keep_rows, s);
      if (rc) return rc;
    }
  } else {
    // the row arrays change places with the workspace's second set; the new set is filled [front | kept | back]
    DBuf *cur[] = {&b->packed[0], &b->packed[1], &b->q[0], &b->q[1], &b->namelen, &b->namecell, &b->name_in_off, &b->tok_bucket, &b->tok_pos};
    DBuf *alt[] = {&w->alt_packed[0], &w->alt_packed[1], &w->alt_q[0], &w->alt_q[1], &w->alt_namelen, &w->alt_namecell, &w->alt_name_in_off,
                   &w->alt_tok_bucket, &w->alt_tok_pos};
    const size_t elem[] = {(size_t)b->stride[0], (size_t)b->stride[1], (size_t)b->qstride[0], (size_t)b->qstride[1], 1, 16, 8, 4, 4};
    for (size_t i = 0; i < sizeof(cur) / sizeof(cur[0]); i++) {
      if (!cur[i]->p) continue;             // (an array this batch does not use: mate 2, names, long names)
      const bool tok = cur[i] == &b->tok_bucket || cur[i] == &b->tok_pos;
      ENSURE(b, *alt[i], std::max<size_t>(cur[i]->cap, tok ? sizeof(u32) * (rows_bound + 1) : 0));  // same capacity: reserve_rows sees one row_cap for both sets
      std::swap(*cur[i], *alt[i]);
    }
    // (the long-name STORE stays: name_in_off holds absolute positions in it, and the names of the rows that leave are only lost space)
    const u64 names_used = b->names_in_used;
    b->N = b->base = b->NP = 0;
    if (have_front) {
      int rc = ingest_piece(b, front, fb, true, used, s);
      if (rc) return rc;
      if (used[0] != fb[0] || (b->nm == 2 && used[1] != fb[1])) { set_err(c, "rewindow: the front text is not whole records"); return SCALCE_ERR_FORMAT; }
    }
    nfront = b->N;
    { int rc = reserve_rows(b, nfront + keep_rows, nfront, s); if (rc) return rc; }
    if (keep_rows) {
      for (size_t i = 0; i < sizeof(cur) / sizeof(cur[0]); i++) {
        if (!cur[i]->p || !alt[i]->p) continue;
        const bool tok = cur[i] == &b->tok_bucket || cur[i] == &b->tok_pos;
        if (tok && !walked) continue;
        if (cur[i] == &b->name_in_off && !names_used) continue;
        HIP_TRY(c, hipMemcpyAsync(cur[i]->as<u8>() + nfront * elem[i], alt[i]->as<u8>() + keep_first * elem[i], keep_rows * elem[i],
                                  hipMemcpyDeviceToDevice, s));
      }
    }
    b->N = nfront + keep_rows;
    b->base = b->N; b->NP = 0;
    b->ingested[0] = true; b->ingested[1] = b->nm == 2;
  }
  const u64 nkept_end = b->N;
  if (bb[0]) {
    int rc = ingest_piece(b, back, bb, true, used, s);
    if (rc) return rc;
    if (used[0] != bb[0] || (b->nm == 2 && used[1] != bb[1])) { set_err(c, "rewindow: the back text is not whole records"); return SCALCE_ERR_FORMAT; }
  }
  // the first walk of the rows that came in (the kept rows keep theirs)
  if (walked && b->tok_bucket.cap >= sizeof(u32) * (b->N + 1) && b->tok_pos.cap >= sizeof(u32) * (b->N + 1)) {
    int rc = first_walk(b, 0, nfront, 0, s);
    if (!rc) rc = first_walk(b, nkept_end, b->N - nkept_end, 0, s);
    if (rc) return rc;
    b->walk_rows = b->N;
    w->walk_owner = b;
  } else {
    b->walk_rows = 0;  // (scalce_batch_tokenize_begin walks every row)
  }
  return SCALCE_OK;
}

// back to rows that lie back to back (before anything is ingested): callers that read q' in input order as one array
// (sharded runs), runs sized for most of HBM (lean)
static void unfuse(scalce_batch *b) {
  if (!b->fused || b->N) return;
  b->fused = false;
  b->qstride[0] = (u32)b->L[0];
}
extern "C" void scalce_batch_set_lean(scalce_batch *b, int lean) {
  if (!b) return;
  b->lean = lean != 0;
  if (b->lean) unfuse(b);
}
extern "C" uint64_t scalce_batch_reruns(const scalce_batch *b) { return b ? b->reruns : 0; }
extern "C" int scalce_batch_set_code_in_place(scalce_batch *b, int on) {
  if (!b) return SCALCE_ERR_ARG;
  if (on && (b->p.no_ac || b->lean)) return SCALCE_ERR_ARG;
  b->code_in_place = on != 0;
  return SCALCE_OK;
}
extern "C" int scalce_batch_set_stream_scratch(scalce_batch *b, int on) {
  if (!b) return SCALCE_ERR_ARG;
  if (on && (b->p.no_ac || b->lean)) return SCALCE_ERR_ARG;  // -A: the stream IS the output (compress.cpp:389-390)
  b->qs_in_ws = on != 0;
  return SCALCE_OK;
}
extern "C" int scalce_batch_set_fused_rows(scalce_batch *b, int on) {
  if (!b) return SCALCE_ERR_ARG;
  if (!on) unfuse(b);
  return (on != 0) == b->fused ? SCALCE_OK : SCALCE_ERR_ARG;
}

// ---- stage 1: quality statistics -------------------------------------------------------------------
extern "C" int scalce_batch_quality(scalce_batch *b, void *stream) {
  if (!b) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  scalce_ctx *c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  StageTimer tm(b, ST_QUALITY, s);
  for (int m = 0; m < b->nm; m++) {
    if (!b->ingested[m]) { set_err(c, "ingest mate %d first", m + 1); return SCALCE_ERR_ARG; }
    if (b->base == 0) {  // counters run over all pieces of the batch
      HIP_TRY(c, hipMemsetAsync(b->freq4[m].p, 0, sizeof(u64) * 512000, s));
      b->tri_expected[m] = 0;
    }
    if (b->p.no_ac) continue;  // statistics are skipped under -A (qualities.cpp:185)
    const u64 n = b->NP * (u64)b->L[m], before = b->base * (u64)b->L[m];
    if (!n) continue;
    const u8 *q = b->q[m].as<u8>() + b->base * (u64)b->qstride[m];  // the piece's first row
    u32 *minmax = b->d_small + 24;  // smallest / largest symbol of the piece
    HIP_TRY(c, hipMemsetAsync(minmax, 0xFF, sizeof(u32), s));
    HIP_TRY(c, hipMemsetAsync(minmax + 1, 0, sizeof(u32), s));
    // the ingest kernel left the range of every tile of the piece's text -- in the WORKSPACE: if another batch that shares it
    // has ingested since, the ranges are that batch's, and the piece's own q' rows are scanned instead
    if (b->mm_valid[m] && b->ws->tile_mm_owner[m] == b) {
      const u32 nt = cdiv(b->text_bytes[m], ING_TILE);
      LAUNCH(tile_minmax_reduce_k, cdiv(nt, 256 * 16) ? cdiv(nt, 256 * 16) : 1, 256, 0, s, b->tile_mm[m].as<u16>(), nt, minmax);
    } else if (b->qstride[m] != (u32)b->L[m]) {
      // (fused rows and no tile ranges -- another batch of the workspace has ingested since: the whole alphabet, more passes)
      static const u32 whole[2] = {0u, 79u};
      HIP_TRY(c, hipMemcpyAsync(minmax, whole, sizeof whole, hipMemcpyHostToDevice, s));
    } else {
      LAUNCH(sym_range_k, 2048, 256, 0, s, q, n, minmax);
    }
    u32 *prev = b->d_small + 20 + 2 * m;  // the two symbols in front of this piece
    LAUNCH(tri_prev_k, 1, 1, 0, s, b->q[m].as<u8>(), (u32)b->L[m], b->qstride[m], before, b->p.qprev[m][0], b->p.qprev[m][1], prev);
    u32 *range = b->d_small + 14;  // {lo, A}: span of the symbols that occur
    LAUNCH(tri_range_k, 1, 1, 0, s, minmax, prev, range);
    unsigned long long *tiles = reinterpret_cast<unsigned long long *>(b->d_small64 + 300);  // one tile counter per pass
    HIP_TRY(c, hipMemsetAsync(tiles, 0, sizeof(u64) * TRI_MAX_PASSES, s));
    for (u32 pass = 0; pass < 3; pass++)  // pass 0, pass 1, and whatever a wide alphabet needs behind them in one launch
      LAUNCH(trigram_pass_k, 256, TRI_THREADS, 0, s, q, n, prev, pass, pass < 2 ? pass + 1 : (u32)TRI_MAX_PASSES, range, b->freq4[m].as<u64>(), tiles,
             (u32)b->L[m], b->qstride[m]);
    {  // one count per symbol with two predecessors (the run's first two have them only when the caller passed qprev)
      const bool p0 = b->p.qprev[m][0] < 80, p1 = b->p.qprev[m][1] < 80;
      const u64 carried = p1 ? (p0 ? 2 : 1) : 0;
      const u64 have = carried + before >= 2 ? 2 : carried + before;
      b->tri_expected[m] += n > 2 - have ? n - (2 - have) : 0;
      u64 *acc = b->d_small64 + 400 + 2 * m;
      HIP_TRY(c, hipMemsetAsync(acc, 0, 2 * sizeof(u64), s));
      LAUNCH(tri_check_k, 64, 256, 0, s, b->freq4[m].as<u64>(), b->tri_expected[m], acc, reinterpret_cast<u32 *>(acc + 1), b->d_err);
    }
  }
  return SCALCE_OK;
}

// the tokenizer walks of a core table too large for the k-mer tables in LDS: occurrences from their starts (tokenize_anchor_k)
static void anchor_args(const scalce_ctx *c, const scalce_batch *b, const u8 *packed, u64 nrec, AnchorArgs &a) {
  memset(&a, 0, sizeof a);
  a.next = reinterpret_cast<const u32 *>(c->d_next); a.outinfo = c->d_outinfo;
  a.bits = c->d_anchor_bits; a.rank = c->d_anchor_rank; a.child = c->d_child_bits; a.K = c->anchor_K; a.idK = c->anchor_idK;
  a.single = c->d_anchor_single;
  a.packed = packed; a.nrec = nrec; a.L = b->L[0]; a.stride = b->stride[0]; a.root_bucket = (u32)c->A.n_buckets;
  a.tok_bucket = b->tok_bucket.as<u32>(); a.tok_pos = b->tok_pos.as<u32>();
}

// pass A of the tokenizer over rows [row0, row0 + n): longest core, its last base, hits at that length, tie flag -> tok_bucket /
// tok_pos at index (row - tok_row0), tok_row0 = the row index 0 of those arrays stands for
static int first_walk(scalce_batch *b, u64 row0, u64 n, u64 tok_row0, hipStream_t s) {
  scalce_ctx *c = b->ctx;
  if (!n) return SCALCE_OK;
  const u8 *packed0 = b->packed[0].as<u8>() + row0 * (u64)b->stride[0];
  TokArgs a;
  a.next = c->d_next; a.outinfo = c->d_outinfo; a.n_states = (u32)c->A.n_states; a.lds_states = (u32)c->tok_lds_states;
  a.packed = packed0; a.nrec = n; a.L = b->L[0]; a.stride = b->stride[0];
  a.root_bucket = (u32)c->A.n_buckets; a.tok_bucket = b->tok_bucket.as<u32>() + (row0 - tok_row0); a.tok_pos = b->tok_pos.as<u32>() + (row0 - tok_row0);
  const size_t sh = (size_t)a.lds_states * 20;
  a.kmer = c->d_kmer; a.id8_first = c->id8_first;
  if (c->anchor_K) {
    AnchorArgs g;
    anchor_args(c, b, packed0, n, g);
    g.tok_bucket = a.tok_bucket; g.tok_pos = a.tok_pos;
    LAUNCH(tokenize_anchor_k<false>, cdiv(n, 256), 256, 0, s, g);
  } else if (c->d_kmer) {
    if (c->kmer_t7_out) LAUNCH(tokenize_kmer_pipe_k<true>, cdiv(n, TOKP_THREADS), TOKP_THREADS, 0, s, a);
    else LAUNCH(tokenize_kmer_pipe_k<false>, cdiv(n, TOKP_THREADS), TOKP_THREADS, 0, s, a);
  }
  else if (a.lds_states) LAUNCH(tokenize_k<true>, cdiv(n, TOK_THREADS), TOK_THREADS, sh, s, a);
  else LAUNCH(tokenize_k<false>, cdiv(n, TOK_THREADS), TOK_THREADS, 0, s, a);
  return SCALCE_OK;
}

// ---- stage 2: tokenize ------------------------------------------------------------------------------
extern "C" int scalce_batch_tokenize_begin(scalce_batch *b, void *stream) {
  if (!b || !b->ingested[0]) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  scalce_ctx *c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  StageTimer tm(b, ST_TOKENIZE, s);
  // the rows not tokenized yet, [tok_base, N): the piece just appended, or everything when the caller deferred it
  b->tok_base = b->tok_done;
  b->tok_n = b->N - b->tok_done;
  const u64 N = b->tok_n;
  const u8 *packed0 = b->packed[0].as<u8>() + b->tok_base * (u64)b->stride[0];
  b->jacobi_iters = 0;
  b->tie_fallback = false;
  b->sweep_no = 0;
  b->tok_open = true;
  const u32 nb1 = (u32)c->A.n_buckets + 1;  // buckets incl. root
  ENSURE(b, b->tok_bucket, sizeof(u32) * (N + 1));
  ENSURE(b, b->tok_pos, sizeof(u32) * (N + 1));
  ENSURE(b, b->tie_index, sizeof(u32) * (N + 1));
  ENSURE(b, b->ev_off, sizeof(u32) * (N + 1));
  ENSURE(b, b->counts, sizeof(u64) * (nb1 + 1));
  if (!b->counts_total.p) {
    ENSURE(b, b->counts_total, sizeof(u64) * (nb1 + 1));
    ENSURE(b, b->prior_buf, sizeof(u64) * (nb1 + 1));
  }
  if (b->tok_base == 0) HIP_TRY(c, hipMemsetAsync(b->counts_total.p, 0, sizeof(u64) * (nb1 + 1), s));
  ENSURE(b, b->seg, 3 * sizeof(u32) * (nb1 + 2));  // segment starts over all events, over the tie events, fixed reads per bucket
  ENSURE(b, b->Gseg, sizeof(u32) * (nb1 + 2));
  ENSURE(b, b->scan_ws, sizeof(u64) * (scan_ws_elems(2 * N + 1024) + 1024));
  u32 *ws32 = b->scan_ws.as<u32>();
  ENSURE(b, b->dirty, 2 * sizeof(u32) * (size_t)(nb1 + 64) + sizeof(u64) * (nb1 + 8));
  if (!N) {
    HIP_TRY(c, hipMemsetAsync(b->counts.p, 0, sizeof(u64) * (nb1 + 1), s));
    b->ntie = b->nev = 0;
    return SCALCE_OK;
  }
  // pass A: every read (unless scalce_batch_chunk_plan / scalce_batch_rewindow have walked exactly these rows already: sharded runs)
  if (!(b->tok_base == 0 && b->walk_rows == N && b->ws->walk_owner == b)) {
    int rc = first_walk(b, b->tok_base, N, b->tok_base, s);
    if (rc) return rc;
  }
  b->walk_rows = 0;  // (the scans below rewrite tok_pos)
  b->ws->walk_owner = nullptr;
  // tie reads: compact, then size the candidate lists by their hit counts
  ENSURE(b, b->tie_read, sizeof(u32) * (N + 1));
  exclusive_scan<u32>(TieFlag{b->tok_pos.as<u32>()}, N,
                      TieCompact{b->tok_pos.as<u32>(), b->tie_index.as<u32>(), b->tie_read.as<u32>()}, ws32, b->d_small, s);
  u32 ntie = 0;
  { int rc = read_u32(b, b->d_small, &ntie, 1, s); if (rc) return rc; }
  // a malformed record (compress.cpp:628-634 exits there) ends the run here, before later stages size anything from its row
  { int rc = check_device_error(b, s); if (rc) { b->tok_open = false; return rc; } }
  b->ntie = ntie;
  ENSURE(b, b->tie_off, sizeof(u32) * (ntie + 2));
  ENSURE(b, b->tie_ncand, sizeof(u32) * (ntie + 2));
  ENSURE(b, b->choice, sizeof(u32) * (ntie + 2));
  u32 ncap = 0;
  if (ntie) {
    exclusive_scan<u32>(TieHits{b->tok_pos.as<u32>(), b->tie_read.as<u32>()}, ntie, StoreTo<u32>{b->tie_off.as<u32>()}, ws32,
                        b->d_small + 1, s);
    int rc = read_u32(b, b->d_small + 1, &ncap, 1, s);
    if (rc) return rc;
  }
  b->ncand_cap = ncap;
  ENSURE(b, b->cand_bucket, sizeof(u32) * (ncap + 2));
  ENSURE(b, b->cand_pos, sizeof(u32) * (ncap + 2));
  ENSURE(b, b->cand_place, sizeof(u32) * (ncap + 2));
  if (ntie) {
    TieArgs a;
    a.next = c->d_next; a.outinfo = c->d_outinfo; a.packed = packed0; a.L = b->L[0]; a.stride = b->stride[0];
    a.ntie = ntie; a.tie_read = b->tie_read.as<u32>(); a.tie_off = b->tie_off.as<u32>(); a.bucket_level = c->d_bucket_level;
    a.tok_bucket = b->tok_bucket.as<u32>(); a.cand_bucket = b->cand_bucket.as<u32>(); a.cand_pos = b->cand_pos.as<u32>();
    a.tie_ncand = b->tie_ncand.as<u32>();
    a.lds_states = (u32)c->tok_lds_states;
    a.kmer = c->d_kmer; a.id8_first = c->id8_first;
    if (c->anchor_K) {
      AnchorArgs g;
      anchor_args(c, b, packed0, N, g);
      g.ntie = ntie; g.tie_read = a.tie_read; g.tie_off = a.tie_off; g.bucket_level = a.bucket_level;
      g.cand_bucket = a.cand_bucket; g.cand_pos = a.cand_pos; g.tie_ncand = a.tie_ncand;
      LAUNCH(tokenize_anchor_k<true>, cdiv(ntie, 256), 256, 0, s, g);
    } else if (c->d_kmer) {
      if (c->kmer_t7_out) LAUNCH(tie_candidates_pipe_k<true>, cdiv(ntie, TOKP_THREADS), TOKP_THREADS, 0, s, a);
      else LAUNCH(tie_candidates_pipe_k<false>, cdiv(ntie, TOKP_THREADS), TOKP_THREADS, 0, s, a);
    }
    else if (a.lds_states) LAUNCH(tie_candidates_k<true>, cdiv(ntie, TOK_THREADS), TOK_THREADS, (size_t)a.lds_states * 20, s, a);
    else LAUNCH(tie_candidates_k<false>, cdiv(ntie, TOK_THREADS), TOK_THREADS, 0, s, a);
    HIP_TRY(c, hipMemsetAsync(b->choice.p, 0, sizeof(u32) * ntie, s));
  }
  // events in read order, stable-sorted by bucket
  exclusive_scan<u32>(EvCount{b->tok_pos.as<u32>(), b->tie_index.as<u32>(), b->tie_ncand.as<u32>()}, N,
                      StoreTo<u32>{b->ev_off.as<u32>()}, ws32, b->d_small + 2, s);
  u32 nev = 0;
  { int rc = read_u32(b, b->d_small + 2, &nev, 1, s); if (rc) return rc; }
  b->nev = nev;
  ENSURE(b, b->ev_sorted, sizeof(u32) * (nev + 2));
  ENSURE(b, b->ev_tmp, sizeof(u32) * (nev + 2));
  ENSURE(b, b->ev_place, sizeof(u32) * (nev + 2));
  ENSURE(b, b->chosen, nev + 64);
  ENSURE(b, b->G, sizeof(u32) * (nev + 2));
  ENSURE(b, b->hist, sizeof(u32) * radix_hist_elems(nev > N ? nev : N));
  ENSURE(b, b->scan_ws, sizeof(u64) * (scan_ws_elems(radix_hist_elems(nev > N ? nev : N)) + scan_ws_elems(nev) + 1024));
  ws32 = b->scan_ws.as<u32>();
  {
    EventArgs a;
    a.nrec = N; a.tok_bucket = b->tok_bucket.as<u32>(); a.tok_pos = b->tok_pos.as<u32>(); a.tie_index = b->tie_index.as<u32>();
    a.tie_off = b->tie_off.as<u32>(); a.tie_ncand = b->tie_ncand.as<u32>(); a.cand_bucket = b->cand_bucket.as<u32>();
    a.ev_off = b->ev_off.as<u32>(); a.ev_bucket = nullptr; a.ev_init = nullptr;  // (the keys carry bucket and flags)
    // the events are sorted by bucket as (key, event) pairs, like the order stage's records (sequential passes; the
    // index-only passes gathered the bucket through the index, and so did the two kernels behind them)
    ENSURE(b, b->key_a, sizeof(u64) * ((size_t)nev + 2));
    ENSURE(b, b->key_b, sizeof(u64) * ((size_t)nev + 2));
    a.ev_key = b->key_a.as<u64>();
    LAUNCH(events_fill_k, cdiv(N, 256), 256, 0, s, a);
  }
  int bits = 1;
  while ((1u << bits) < nb1 && bits < 31) bits++;
  const u32 *src = nullptr;  // identity
  u32 *dst = b->ev_sorted.as<u32>(), *alt = b->ev_tmp.as<u32>();
  u64 *ka = b->key_a.as<u64>(), *kb = b->key_b.as<u64>();
  for (int sh = 2; sh < 2 + bits; sh += 8) {  // bits 0, 1 (initial flag, tie bit) ride along
    radix_pass_kv(ka, src, kb, dst, nev, (u32)sh, b->hist.as<u32>(), ws32, s);
    src = dst;
    u32 *t = dst; dst = alt; alt = t;
    u64 *tk = ka; ka = kb; kb = tk;
  }
  const u32 *sorted = src;
  // compact view of the tie-candidate events (see events_place_keys_k): what the sweeps work on
  u32 *cidx = dst;  // the ping-pong buffer the sort no longer needs
  exclusive_scan<u32>(TieBitOfKey{ka}, nev, StoreTo<u32>{cidx}, ws32, b->d_small + 3, s);
  u32 ntev = 0;
  { int rc = read_u32(b, b->d_small + 3, &ntev, 1, s); if (rc) return rc; }
  b->ntev = ntev;
  ENSURE(b, b->cand_fixed, sizeof(u32) * (ncap + 2));
  u32 *seg_all = b->seg.as<u32>(), *seg_t = seg_all + (nb1 + 2), *fixed_total = seg_t + (nb1 + 2);
  LAUNCH(events_place_keys_k, cdiv(nev, 256), 256, 0, s, nev, sorted, ka, cidx, b->ev_place.as<u32>(), b->chosen.as<u8>());
  LAUNCH(events_segments_keys_k, cdiv((u64)nev + 1, 256), 256, 0, s, nev, ka, nb1, seg_all);
  LAUNCH(events_compact_segments_k, cdiv(nb1 + 1, 256), 256, 0, s, nb1, seg_all, cidx, nev, ntev, seg_t, fixed_total);
  if (ntie)
    LAUNCH(tie_place_k, cdiv(ntie, 256), 256, 0, s, ntie, b->tie_read.as<u32>(), b->tie_off.as<u32>(), b->tie_ncand.as<u32>(),
           b->ev_off.as<u32>(), b->ev_place.as<u32>(), cidx, b->cand_bucket.as<u32>(), seg_all, seg_t, b->cand_place.as<u32>(),
           b->cand_fixed.as<u32>());
  // first prefix sums (per bucket) and counts; the sweeps follow (scalce_batch_tokenize_sweep)
  b->dirty_cur = 0;
  HIP_TRY(c, hipMemsetAsync(b->dirty.p, 0, sizeof(u32) * nb1, s));  // first sweep: every bucket moved "before read 0"
  HIP_TRY(c, hipMemsetAsync(b->dirty.as<u32>() + 2 * (size_t)(nb1 + 64), 0, sizeof(u64) * nb1, s));  // prior seen so far
  LAUNCH(seg_rescan_k, nb1, 256, 0, s, nb1, seg_t, b->dirty.as<u32>(), b->chosen.as<u8>(), b->G.as<u32>(), fixed_total, b->counts.as<u64>());
  return SCALCE_OK;
}

// One Jacobi sweep, enqueued only: decisions of the tie reads against the current counts, then new prefix sums and
// per-bucket counts for the buckets whose flags moved (seg_rescan_k looks at the dirty marks itself: nothing moved,
// nothing to do).  flag[0] becomes 1 if any decision of this shard moved.
static int tokenize_sweep_enqueue(scalce_batch *b, const uint64_t *d_prior, u32 *flag, hipStream_t s) {
  scalce_ctx *c = b->ctx;
  const u32 nb1 = (u32)c->A.n_buckets + 1, ntie = b->ntie;
  if (b->tok_base) {  // reads of this batch's earlier pieces count as well (bin_size is cumulative, reads.cpp:246)
    if (d_prior) {
      LAUNCH(add_counts_k, cdiv(nb1, 256), 256, 0, s, nb1, reinterpret_cast<const u64 *>(d_prior), b->counts_total.as<u64>(), b->prior_buf.as<u64>());
      d_prior = b->prior_buf.as<uint64_t>();
    } else {
      d_prior = b->counts_total.as<uint64_t>();
    }
  }
  u32 *d0 = b->dirty.as<u32>(), *d1 = d0 + nb1 + 64;
  u32 *dirty_in = b->dirty_cur ? d1 : d0, *dirty_out = b->dirty_cur ? d0 : d1;
  u64 *prior_seen = reinterpret_cast<u64 *>(d0 + 2 * (size_t)(nb1 + 64));
  if (d_prior) LAUNCH(prior_dirty_k, cdiv(nb1, 256), 256, 0, s, nb1, reinterpret_cast<const u64 *>(d_prior), prior_seen, dirty_in);
  HIP_TRY(c, hipMemsetAsync(flag, 0, sizeof(u32), s));
  u32 *G = b->G.as<u32>();
  JacobiArgs a;
  a.ntie = ntie; a.tie_read = b->tie_read.as<u32>(); a.tie_off = b->tie_off.as<u32>(); a.tie_ncand = b->tie_ncand.as<u32>();
  a.cand_bucket = b->cand_bucket.as<u32>(); a.cand_place = b->cand_place.as<u32>(); a.G = G; a.fixed_before = b->cand_fixed.as<u32>();
  a.prior = reinterpret_cast<const u64 *>(d_prior); a.choice = b->choice.as<u32>(); a.chosen = b->chosen.as<u8>();
  a.changed = flag;
  a.dirty_in = dirty_in; a.dirty_out = dirty_out;
  {
    a.coarse = b->sweep_no < 8u ? 1u : 0u;
    b->sweep_no++;
  }
  HIP_TRY(c, hipMemsetAsync(dirty_out, 0xFF, sizeof(u32) * nb1, s));
  LAUNCH(jacobi_k, cdiv(ntie, 256), 256, 0, s, a);
  b->dirty_cur ^= 1;
  LAUNCH(seg_rescan_k, nb1, 256, 0, s, nb1, b->seg.as<u32>() + (nb1 + 2), dirty_out, b->chosen.as<u8>(), G, b->seg.as<u32>() + 2 * (nb1 + 2),
         b->counts.as<u64>());
  return SCALCE_OK;
}

// One sweep with the given cross-shard prior counts (SCALCE_OUT_BUCKET_COUNTS is current when it returns).
// *changed = 1 if any decision of THIS shard moved.
// The tie reads decided in input order by one wavefront (tie_sequential_k): what scalce_batch_tokenize falls back to when
// the sweeps have not reached their fixed point after tie_max_sweeps() of them.  Leaves choice / chosen / G / counts as
// the converged sweeps would.
static u32 tie_max_sweeps() {
  const char *e = getenv("SCALCE_TIE_MAX_SWEEPS");  // (tests lower it to drive the fallback on ordinary input)
  const int v = e ? atoi(e) : 256;
  return (u32)(v < 1 ? 1 : v);
}
static int tokenize_sequential(scalce_batch *b, const uint64_t *d_prior, hipStream_t s) {
  scalce_ctx *c = b->ctx;
  const u32 nb1 = (u32)c->A.n_buckets + 1, ntie = b->ntie;
  if (b->tok_base) {
    if (d_prior) {
      LAUNCH(add_counts_k, cdiv(nb1, 256), 256, 0, s, nb1, reinterpret_cast<const u64 *>(d_prior), b->counts_total.as<u64>(), b->prior_buf.as<u64>());
      d_prior = b->prior_buf.as<uint64_t>();
    } else {
      d_prior = b->counts_total.as<uint64_t>();
    }
  }
  ENSURE(b, b->Gseg, sizeof(u32) * (nb1 + 2));
  TieSeqArgs a;
  a.ntie = ntie; a.nb1 = nb1; a.tie_off = b->tie_off.as<u32>(); a.tie_ncand = b->tie_ncand.as<u32>(); a.cand_bucket = b->cand_bucket.as<u32>();
  a.fixed_before = b->cand_fixed.as<u32>(); a.prior = reinterpret_cast<const u64 *>(d_prior); a.choice = b->choice.as<u32>();
  a.tiecount = b->Gseg.as<u32>();
  const size_t lds = (size_t)nb1 * 4;
  a.lds_counters = lds <= 100 * 1024 ? 1u : 0u;
  if (!a.lds_counters) HIP_TRY(c, hipMemsetAsync(a.tiecount, 0, sizeof(u32) * nb1, s));
  if (a.lds_counters && lds > 48 * 1024)
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(tie_sequential_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  LAUNCH(tie_sequential_k, 1, 64, a.lds_counters ? lds : 0, s, a);
  HIP_TRY(c, hipMemsetAsync(b->chosen.p, 0, b->nev + 64, s));
  LAUNCH(chosen_from_choice_k, cdiv(ntie, 256), 256, 0, s, ntie, b->tie_off.as<u32>(), b->choice.as<u32>(), b->cand_place.as<u32>(), b->chosen.as<u8>());
  u32 *d0 = b->dirty.as<u32>();
  HIP_TRY(c, hipMemsetAsync(d0, 0, sizeof(u32) * nb1, s));  // every bucket: new prefix sums and counts
  LAUNCH(seg_rescan_k, nb1, 256, 0, s, nb1, b->seg.as<u32>() + (nb1 + 2), d0, b->chosen.as<u8>(), b->G.as<u32>(), b->seg.as<u32>() + 2 * (nb1 + 2),
         b->counts.as<u64>());
  b->tie_fallback = true;
  return SCALCE_OK;
}

// Several sweeps against the same prior counts with ONE look at their flags (a sweep behind the local fixed point changes
// nothing and costs next to nothing; a host round trip per sweep leaves the stream idle).  *changed = 1 if any moved.
extern "C" int scalce_batch_tokenize_end(scalce_batch *b, void *stream) {
  if (!b || !b->tok_open) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  scalce_ctx *c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  StageTimer tm(b, ST_TOKENIZE, s);
  b->tok_open = false;
  const u64 N = b->tok_n;
  b->tok_done = b->tok_base + b->tok_n;
  const u32 nb1 = (u32)c->A.n_buckets + 1;
  // reads per bucket over all pieces so far: what the next piece's tie-break starts from, and what the emit stage lays out
  LAUNCH(add_counts_k, cdiv(nb1, 256), 256, 0, s, nb1, b->counts.as<u64>(), b->counts_total.as<u64>(), b->counts_total.as<u64>());
  if (!N) return SCALCE_OK;
  FinalizeArgs a;
  a.nrec = N; a.tok_bucket = b->tok_bucket.as<u32>(); a.tok_pos = b->tok_pos.as<u32>(); a.tie_index = b->tie_index.as<u32>();
  a.tie_off = b->tie_off.as<u32>(); a.choice = b->choice.as<u32>(); a.cand_bucket = b->cand_bucket.as<u32>();
  a.cand_pos = b->cand_pos.as<u32>(); a.bucket_pattern = c->d_bucket_pattern; a.root_bucket = (u32)c->A.n_buckets;
  a.bucket = b->bucket.as<u32>() + b->tok_base; a.end = b->endv.as<u16>() + b->tok_base; a.tokens = b->tokens.as<int32_t>() + 2 * b->tok_base;
  LAUNCH(finalize_k, cdiv(N, 256), 256, 0, s, a);
  return SCALCE_OK;
}

// The tie-break of a batch on its own (fixed prior counts), window by window: see tie_window_sweep_k.  *settled = false
// when the sweeps allowed (tie_max_sweeps() per window) did not get through: the caller decides in input order instead.
static u32 tie_window_reads() {
  const char *e = getenv("SCALCE_TIE_WINDOW");  // tie reads per window; 0 = the global sweeps (jacobi_k)
  const long v = e ? atol(e) : 196608;
  return (u32)(v < 0 ? 0 : v > (1l << 30) ? (1l << 30) : v);
}
static int tokenize_windows(scalce_batch *b, const uint64_t *d_prior, bool *settled, hipStream_t s) {
  scalce_ctx *c = b->ctx;
  const u32 nb1 = (u32)c->A.n_buckets + 1, ntie = b->ntie, ncap = b->ncand_cap, ntev = b->ntev;
  *settled = false;
  if (b->tok_base) {  // reads of this batch's earlier pieces count as well (bin_size is cumulative, reads.cpp:246)
    if (d_prior) {
      LAUNCH(add_counts_k, cdiv(nb1, 256), 256, 0, s, nb1, reinterpret_cast<const u64 *>(d_prior), b->counts_total.as<u64>(), b->prior_buf.as<u64>());
      d_prior = b->prior_buf.as<uint64_t>();
    } else {
      d_prior = b->counts_total.as<uint64_t>();
    }
  }
  u32 W = tie_window_reads();
  const u64 max_cells = 64ull << 20;  // (a million-core table: fewer, larger windows)
  if ((u64)cdiv(ntie, W) * nb1 > max_cells) W = (u32)cdiv(ntie, max_cells / nb1 ? max_cells / nb1 : 1);
  const u32 nwin = cdiv(ntie, W);
  const u64 ncells = (u64)nwin * nb1;
  const u64 nwords = ((u64)ntev >> 6) + 4;
  ENSURE(b, b->tw_cells, sizeof(u32) * (3 * ncells + 8));
  ENSURE(b, b->tw_cand, sizeof(u32) * (2 * (u64)ncap + 8));
  ENSURE(b, b->tw_bits, (sizeof(u64) + sizeof(u32)) * nwords);
  ENSURE(b, b->tw_base, sizeof(u32) * (2 * (u64)nb1 + 64));
  ENSURE(b, b->scan_ws, sizeof(u64) * (scan_ws_elems(ncells) + 1024));
  u32 *first_delta = b->tw_cells.as<u32>(), *cell_end = first_delta + ncells, *cellstart = cell_end + ncells + 2;  // cellstart[ncells] = all candidates
  u32 *wpos = b->tw_cand.as<u32>(), *rs = wpos + ncap + 2;
  u64 *bits = b->tw_bits.as<u64>();
  u32 *P64 = reinterpret_cast<u32 *>(bits + nwords);
  u32 *base = b->tw_base.as<u32>();
  TieWinState *st = reinterpret_cast<TieWinState *>(base + nb1 + 4);
  const u32 *fixed_total = b->seg.as<u32>() + 2 * (nb1 + 2);
  u32 *key_c = b->G.as<u32>();  // (the global sweeps' prefix sums: not in use here)
  HIP_TRY(c, hipMemsetAsync(first_delta, 0, sizeof(u32) * 2 * ncells, s));  // empty cells: first = end = 0
  HIP_TRY(c, hipMemsetAsync(bits, 0, (sizeof(u64) + sizeof(u32)) * nwords, s));
  HIP_TRY(c, hipMemsetAsync(base, 0, sizeof(u32) * (2 * (u64)nb1 + 64), s));  // (and the state behind it)
  HIP_TRY(c, hipMemsetAsync(b->choice.p, 0xFF, sizeof(u32) * ntie, s));   // nobody has chosen yet
  LAUNCH(tw_key_k, cdiv(ntie, 256), 256, 0, s, ntie, W, nb1, b->tie_off.as<u32>(), b->tie_ncand.as<u32>(), b->cand_bucket.as<u32>(),
         b->cand_place.as<u32>(), key_c);
  if (ntev) LAUNCH(tw_heads_k, cdiv(ntev, 256), 256, 0, s, ntev, key_c, first_delta, cell_end);
  exclusive_scan<u32>(CellCount{first_delta, cell_end}, ncells, StoreTo<u32>{cellstart}, b->scan_ws.as<u32>(), cellstart + ncells, s);
  LAUNCH(tw_delta_k, cdiv(ncells, 256), 256, 0, s, ncells, cellstart, first_delta);
  LAUNCH(tw_cand_k, cdiv(ntie, 256), 256, 0, s, ntie, W, nb1, b->tie_off.as<u32>(), b->tie_ncand.as<u32>(), b->cand_bucket.as<u32>(),
         b->cand_place.as<u32>(), cellstart, first_delta, wpos, rs);
  TieWinArgs a;
  a.ntie = ntie; a.W = W; a.tie_off = b->tie_off.as<u32>(); a.tie_ncand = b->tie_ncand.as<u32>(); a.cand_bucket = b->cand_bucket.as<u32>();
  a.fixed_before = b->cand_fixed.as<u32>(); a.wpos = wpos; a.rs = rs; a.prior = reinterpret_cast<const u64 *>(d_prior); a.base = base;
  a.P64 = P64; a.bits = bits; a.bits32 = reinterpret_cast<u32 *>(bits); a.choice = b->choice.as<u32>(); a.st = st;
  // A window settles in a handful of sweeps when sweeping works at all (the last one moves nothing), so the sweeps go out in
  // batches sized for the windows still open, and the host looks at the device's state once per batch.
  const u64 budget = (u64)tie_max_sweeps() * nwin;
  // one launch per sweep (tie_window_fused_k) when a window's words and their ranks fit a workgroup's LDS
  // (SCALCE_TIE_WINDOW=<reads>:two_launches: test hook -- the path of windows that do not fit)
  const char *tw_env = getenv("SCALCE_TIE_WINDOW");
  if (!(tw_env && strstr(tw_env, ":two_launches"))) {
    TieFusedState *fs = reinterpret_cast<TieFusedState *>(base + 2 * (u64)nb1 + 16);
    u32 *maxw_d = base + 2 * (u64)nb1 + 32;
    LAUNCH(tw_maxwin_k, cdiv(nwin, 256), 256, 0, s, nwin, nb1, cellstart, maxw_d);
    u32 maxw = 0;
    { int rc = read_u32(b, maxw_d, &maxw, 1, s); if (rc) return rc; }
    const size_t lds = (size_t)(maxw + 2) * 12;
    if (lds <= 140 * 1024) {
      if (lds > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(tie_window_fused_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      static const u32 one = 1;
      HIP_TRY(c, hipMemcpyAsync(&fs->changed[2], &one, sizeof(u32), hipMemcpyHostToDevice, s));  // launch 0: "something moved": sweep window 0
      TieFusedArgs g;
      g.a = a; g.nwin = nwin; g.nb1 = nb1; g.lds_words = maxw + 2; g.cellstart = cellstart; g.base2 = base; g.fs = fs;
      const u32 fgrid = cdiv(W < ntie ? W : ntie, TWF_THREADS);
      u32 n = 0;
      TieFusedState h;
      memset(&h, 0, sizeof h);
      for (;;) {
        if (h.sweeps >= budget) { b->jacobi_iters = h.sweeps; return SCALCE_OK; }  // not settled
        const u32 win_now = n ? h.window[(n - 1) & 1] : 0u;
        u64 batch = 4ull * (nwin - win_now) + 4;
        if (batch > 256) batch = 256;
        if (batch > budget - h.sweeps) batch = budget - h.sweeps;
        for (u64 i = 0; i < batch; i++) {
          g.n = n++;
          LAUNCH(tie_window_fused_k, fgrid, TWF_THREADS, lds, s, g);
        }
        int rc = read_u32(b, reinterpret_cast<const u32 *>(fs), reinterpret_cast<u32 *>(&h), 8, s);
        if (rc) return rc;
        if (h.window[(n - 1) & 1] >= nwin) break;
      }
      b->jacobi_iters = h.sweeps;
      LAUNCH(twf_counts_k, cdiv(nb1, 256), 256, 0, s, nb1, fixed_total, base, fs, n - 1, b->counts.as<u64>());
      *settled = true;
      return SCALCE_OK;
    }
  }
  const u32 grid = cdiv(W < ntie ? W : ntie, 256);
  TieWinState h{0, 0, 0, 0};
  while (!h.finished) {
    if (h.sweeps >= budget) { b->jacobi_iters = h.sweeps; return SCALCE_OK; }  // not settled
    u64 batch = 4ull * (nwin - h.window) + 4;
    if (batch > 256) batch = 256;
    if (batch > budget - h.sweeps) batch = budget - h.sweeps;
    for (u64 i = 0; i < batch; i++) {
      LAUNCH(tie_window_sweep_k, grid, 256, 0, s, a);
      LAUNCH(tie_window_tail_k, 1, 1024, 0, s, st, nwin, nb1, cellstart, bits, P64, base);
    }
    int rc = read_u32(b, reinterpret_cast<const u32 *>(st), reinterpret_cast<u32 *>(&h), 4, s);
    if (rc) return rc;
  }
  b->jacobi_iters = h.sweeps;
  LAUNCH(tw_counts_k, cdiv(nb1, 256), 256, 0, s, nb1, fixed_total, base, b->counts.as<u64>());
  *settled = true;
  return SCALCE_OK;
}

extern "C" int scalce_batch_tokenize(scalce_batch *b, const uint64_t *d_prior, void *stream) {
  int rc = scalce_batch_tokenize_begin(b, stream);
  if (rc) return rc;
  return scalce_batch_tokenize_settle(b, d_prior, stream);
}

// The rest of scalce_batch_tokenize behind _begin: the tie-break of this batch on its own (fixed prior counts), then _end.
// (A caller may put other work of the shard beside it; the quality statistics on a second stream were tried twice: no gain.)
extern "C" int scalce_batch_tokenize_settle(scalce_batch *b, const uint64_t *d_prior, void *stream) {
  if (!b || !b->tok_open) return SCALCE_ERR_ARG;
  HIP_TRY(b->ctx, hipSetDevice(b->ctx->device));
  int rc;
  if (b->tok_n && b->ntie && tie_window_reads()) {
    hipStream_t ws = (hipStream_t)stream;
    {
      StageTimer tm(b, ST_TOKENIZE, ws);
      bool settled = false;
      if ((rc = tokenize_windows(b, d_prior, &settled, ws))) return rc;
      if (!settled && (rc = tokenize_sequential(b, d_prior, ws))) return rc;
    }
    return scalce_batch_tokenize_end(b, stream);
  }
  // Sweeps go out four at a time and the host looks at their flags once per batch: a sweep after the fixed point changes
  // nothing (and costs next to nothing), while a round trip per sweep left the stream idle 47 times per shard.
  hipStream_t s = (hipStream_t)stream;
  constexpr int SWEEPS_PER_LOOK = 4;
  for (bool done = !(b->tok_n && b->ntie); !done;) {
    StageTimer tm(b, ST_TOKENIZE, s);
    u32 *flags = b->d_small + 32;
    for (int i = 0; i < SWEEPS_PER_LOOK; i++)
      if ((rc = tokenize_sweep_enqueue(b, d_prior, flags + i, s))) return rc;
    u32 ch[SWEEPS_PER_LOOK];
    if ((rc = read_u32(b, flags, ch, SWEEPS_PER_LOOK, s))) return rc;
    for (int i = 0; i < SWEEPS_PER_LOOK && !done; i++) {
      b->jacobi_iters++;  // sweeps up to and including the first one that moved nothing, as one at a time would count
      if (!ch[i]) done = true;
    }
    if (!done && b->jacobi_iters >= tie_max_sweeps()) {  // worst cases are quadratic in sweeps: decide in input order instead
      if ((rc = tokenize_sequential(b, d_prior, s))) return rc;
      done = true;
    }
  }
  return scalce_batch_tokenize_end(b, stream);
}

// Sharded runs: where the -B rule (compress.cpp:702-715) cuts this rank's rows, given the bytes already in the chunk that
// is open when they begin.  Record sizes need every row's core LENGTH only (the candidates of a tie are equally long), so
// this runs before the tie-break: the first scan of the tokenizer over all rows, a prefix sum of the sizes, the cuts.
extern "C" int scalce_batch_chunk_plan(scalce_batch *b, uint64_t carry_in, uint64_t *cuts_host, uint32_t cap, uint32_t *ncuts,
                                       uint64_t *carry_out, void *stream) {
  if (!b || !ncuts || !carry_out || (cap && !cuts_host) || !b->ingested[0]) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  scalce_ctx *c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  StageTimer tm(b, ST_TOKENIZE, s);
  const u64 N = b->N;
  *ncuts = 0;
  *carry_out = carry_in;
  if (!N) return SCALCE_OK;
  ENSURE(b, b->tok_bucket, sizeof(u32) * (N + 1));
  ENSURE(b, b->tok_pos, sizeof(u32) * (N + 1));
  if (b->S_rows != N && !(b->tok_done == 0 && b->walk_rows == N && b->ws->walk_owner == b)) {  // (a second call with another carry_in only redoes the cuts)
    int rc = first_walk(b, 0, N, 0, s);
    if (rc) return rc;
    b->walk_rows = b->tok_done == 0 ? N : 0;  // (tok_bucket / tok_pos are indexed from the first row not tokenized yet)
    b->ws->walk_owner = b;
  }
  ENSURE(b, b->S, sizeof(u64) * (N + 2));
  ENSURE(b, b->scan_ws, sizeof(u64) * (scan_ws_elems(N + 1) + 1024));
  ENSURE(b, b->chunk_start, sizeof(u64) * (cap + 8));
  u64 *S = b->S.as<u64>();
  if (b->S_rows != N) {
    RecSize rs{b->tok_bucket.as<u32>(), c->d_bucket_level, b->namelen.as<u8>(), b->L[0], b->L[1], b->p.paired, b->p.use_names, 1};
    exclusive_scan<u64>(rs, N, StoreTo<u64>{S}, b->scan_ws.as<u64>(), S + N, s);
    b->S_rows = N;
  }
  LAUNCH(chunk_cuts_k, 1, 1, 0, s, S, N, (u64)b->p.bucket_set_size, (u64)carry_in, cap, b->chunk_start.as<u64>(), b->d_small + 8, b->d_small64 + 7);
  u32 n = 0;
  { int rc = read_u32(b, b->d_small + 8, &n, 1, s); if (rc) return rc; }
  { u64 co = 0; int rc = read_u64(b, b->d_small64 + 7, &co, 1, s); if (rc) return rc; *carry_out = co; }
  if (n) {  // (on the caller's stream: a blocking copy would go through the null stream, which does not wait for `s`)
    HIP_TRY(c, hipMemcpyAsync(cuts_host, b->chunk_start.p, sizeof(u64) * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
  }
  *ncuts = n;
  return SCALCE_OK;
}

extern "C" int scalce_batch_text_offset(scalce_batch *b, int mate, uint64_t row, uint64_t *offset, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!b || !offset || mate < 0 || mate >= b->nm || row > b->NP) return SCALCE_ERR_ARG;
  *offset = 0;
  if (!row) return SCALCE_OK;
  HIP_TRY(b->ctx, hipSetDevice(b->ctx->device));
  // from the per-tile newline counts of the piece (its text must still be where it was): no line index is built for this
  const u64 nbytes = b->text_bytes[mate];
  const u32 ntiles = cdiv(nbytes, IDX_TILE);
  if (!ntiles || !b->piece_text[mate]) { set_err(b->ctx, "no piece ingested"); return SCALCE_ERR_ARG; }
  // on the caller's stream, behind the ingest that produced the tile counts, and in a word of its own (slot 7 belongs to
  // scalce_batch_chunk_plan's carry)
  u64 *d_out = b->d_small64 + 10;
  LAUNCH(line_offset_k, 1, 64, 0, s, b->piece_text[mate], nbytes, b->tile[mate].as<u64>(), ntiles, (u64)(4 * row), d_out);
  u64 v = 0;
  { int rc = read_u64(b, d_out, &v, 1, s); if (rc) return rc; }
  *offset = v;
  return SCALCE_OK;
}

extern "C" int scalce_batch_set_chunks(scalce_batch *b, const uint64_t *starts, uint32_t n) {
  if (!b || (n && !starts) || n > 4096) return SCALCE_ERR_ARG;
  b->explicit_chunks.assign(starts, starts + n);
  return SCALCE_OK;
}

// ---- stage 3: order ----------------------------------------------------------------------------------
extern "C" int scalce_batch_order(scalce_batch *b, void *stream) {
  if (!b) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  scalce_ctx *c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  StageTimer tm(b, ST_ORDER, s);
  const u64 N = b->N;
  const u32 nb1 = (u32)c->A.n_buckets + 1;
  ENSURE(b, b->perm_a, sizeof(u32) * (N + 2));
  ENSURE(b, b->perm_b, sizeof(u32) * (N + 2));
  ENSURE(b, b->hist, sizeof(u32) * radix_hist_elems(N));
  ENSURE(b, b->scan_ws, sizeof(u64) * (scan_ws_elems(radix_hist_elems(N)) + scan_ws_elems(N + 1) + 1024));
  b->perm = b->perm_a.as<u32>();
  b->sorted_keys = nullptr;
  b->nchunks = 1;
  if (!N) return SCALCE_OK;
  u32 *ws32 = b->scan_ws.as<u32>();
  // spill chunks: given explicitly (sharded runs: one chunk per shard) or by the -B rule
  if (!b->explicit_chunks.empty()) {
    const u32 nc = (u32)b->explicit_chunks.size();
    ENSURE(b, b->chunk, sizeof(u32) * (N + 2));
    ENSURE(b, b->chunk_start, sizeof(u64) * (nc + 2));
    std::vector<u64> cs(b->explicit_chunks.begin(), b->explicit_chunks.end());
    cs.push_back(N);
    HIP_TRY(c, hipMemcpyAsync(b->chunk_start.p, cs.data(), sizeof(u64) * cs.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(b->d_small + 8, &nc, sizeof(u32), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    b->nchunks = nc;
    if (nc > 1) LAUNCH(chunk_assign_k, cdiv(N, 256), 256, 0, s, N, b->chunk_start.as<u64>(), b->d_small + 8, b->chunk.as<u32>());
  } else if (b->p.bucket_set_size) {
    ENSURE(b, b->S, sizeof(u64) * (N + 2));
    b->S_rows = ~0ull;
    ENSURE(b, b->chunk, sizeof(u32) * (N + 2));
    const u32 max_chunks = 4096;
    ENSURE(b, b->chunk_start, sizeof(u64) * (max_chunks + 2));
    RecSize rs{b->bucket.as<u32>(), c->d_bucket_level, b->namelen.as<u8>(), b->L[0], b->L[1], b->p.paired, b->p.use_names, 1};
    u64 *S = b->S.as<u64>();
    exclusive_scan<u64>(rs, N, StoreTo<u64>{S}, b->scan_ws.as<u64>(), S + N, s);
    LAUNCH(chunk_bounds_k, 1, 1, 0, s, S, N, (u64)b->p.bucket_set_size, max_chunks, b->chunk_start.as<u64>(), b->d_small + 8);
    int rc = read_u32(b, b->d_small + 8, &b->nchunks, 1, s);
    if (rc) return rc;
    if (b->nchunks > 1)
      LAUNCH(chunk_assign_k, cdiv(N, 256), 256, 0, s, N, b->chunk_start.as<u64>(), b->d_small + 8, b->chunk.as<u32>());
  }
  const u32 *src = nullptr;
  u32 *dst = b->perm_a.as<u32>(), *alt = b->perm_b.as<u32>();
  auto flip = [&]() { src = dst; u32 *t = dst; dst = alt; alt = t; };
  const int ndig = (b->L[0] + 3) / 4;
  const bool two_phase = getenv("SCALCE_ORDER_SINGLE_PHASE") == nullptr;  // test hook: all digits in one go
  const int ndig1 = two_phase ? (ndig < PREFIX_DIGITS ? ndig : PREFIX_DIGITS) : ndig;
  const u32 *chunk_or_null = b->nchunks > 1 ? b->chunk.as<u32>() : nullptr;
  int bits = 1;
  while ((1u << bits) < nb1 && bits < 31) bits++;
  int cbits = 0;
  while (b->nchunks > 1 && (1u << cbits) < b->nchunks) cbits++;
  // phase 1 on (key, read) pairs when bucket | chunk | 32 prefix bits fit 64 bits (always, short of millions of cores
  // together with thousands of chunks): every pass then reads and writes sequentially.  The index-only passes below
  // gather a digit through the index in every pass: 8 GB of sector fetches per pass at 50 M reads, and the scattered
  // accesses are what slows a coder launch running beside the order stage most (tools/coder_beside.py).
  const bool by_pairs = two_phase && PREFIX_BITS + cbits + bits <= 64;
  u64 *sorted_keys = nullptr;
  const u32 end_bits = (16 + PREFIX_BITS + cbits + bits <= 64) ? 16u : 0u;
  if (by_pairs) {
    ENSURE(b, b->key_a, sizeof(u64) * (N + 2));
    ENSURE(b, b->key_b, sizeof(u64) * (N + 2));
    u64 *ka = b->key_a.as<u64>(), *kb = b->key_b.as<u64>();
    LAUNCH(order_keys_k, cdiv(N, 256), 256, 0, s, (u32)N, b->bucket.as<u32>(), chunk_or_null, (u32)cbits, b->packed[0].as<u8>(),
           b->endv.as<u16>(), b->L[0], b->stride[0], ndig1, end_bits, ka);
    for (int sh = (int)end_bits; sh < (int)end_bits + PREFIX_BITS + cbits + bits; sh += 8) {
      radix_pass_kv(ka, src, kb, dst, (u32)N, (u32)sh, b->hist.as<u32>(), ws32, s);
      flip();
      u64 *t = ka; ka = kb; kb = t;
    }
    sorted_keys = ka;
    b->sorted_keys = ka;
    b->key_end_bits = end_bits;
    b->key_bucket_shift = end_bits + PREFIX_BITS + (u32)cbits;
    b->key_bucket_mask = (1u << bits) - 1;
  } else {
    // phase 1: first ndig1 key digits (least significant first), then chunk, then bucket
    for (int d = ndig1 - 1; d >= 0; d--) {
      radix_pass(src, dst, (u32)N, KeyDigit{b->packed[0].as<u8>(), b->endv.as<u16>(), b->L[0], b->stride[0], d}, b->hist.as<u32>(),
                 ws32, s);
      flip();
    }
    if (b->nchunks > 1)
      for (int sh = 0; (1u << sh) < b->nchunks; sh += 8) {
        radix_pass(src, dst, (u32)N, DigitOfArray{b->chunk.as<u32>(), sh}, b->hist.as<u32>(), ws32, s);
        flip();
      }
    for (int sh = 0; sh < bits; sh += 8) {
      radix_pass(src, dst, (u32)N, DigitOfArray{b->bucket.as<u32>(), sh}, b->hist.as<u32>(), ws32, s);
      flip();
    }
  }
  u32 *perm1 = const_cast<u32 *>(src);
  b->order_run_members = 0;
  if (ndig1 < ndig) {
    // phase 2: records that still tie on (bucket, chunk, prefix) are sorted on the remaining digits, run by run
    ENSURE(b, b->run_head, N + 64);
    ENSURE(b, b->run_hcount, sizeof(u32) * (N + 2));
    ENSURE(b, b->run_rank, sizeof(u32) * (N + 2));
    ENSURE(b, b->runid, sizeof(u32) * (N + 2));
    RunArgs ra{(u32)N, perm1, b->bucket.as<u32>(), chunk_or_null, b->packed[0].as<u8>(), b->endv.as<u16>(), b->L[0], b->stride[0], ndig1};
    u8 *head = b->run_head.as<u8>();
    if (sorted_keys) LAUNCH(run_heads_keys_k, cdiv(N, 256), 256, 0, s, (u32)N, sorted_keys, end_bits, head);
    else LAUNCH(run_heads_k, cdiv(N, 256), 256, 0, s, ra, head);
    exclusive_scan<u32>(LoadAs<u8, u32>{head}, N, StoreTo<u32>{b->run_hcount.as<u32>()}, ws32, (u32 *)nullptr, s);
    exclusive_scan<u32>(RunMember{head, (u32)N}, N, StoreTo<u32>{b->run_rank.as<u32>()}, ws32, b->d_small + 9, s);
    u32 M = 0;
    { int rc = read_u32(b, b->d_small + 9, &M, 1, s); if (rc) return rc; }
    b->order_run_members = M;
    if (M) {
      ENSURE(b, b->run_items_a, sizeof(u32) * (M + 2));
      ENSURE(b, b->run_items_b, sizeof(u32) * (M + 2));
      ENSURE(b, b->run_pos, sizeof(u32) * (M + 2));
      LAUNCH(run_compact_k, cdiv(N, 256), 256, 0, s, (u32)N, head, b->run_rank.as<u32>(), b->run_hcount.as<u32>(), perm1,
             b->run_items_a.as<u32>(), b->run_pos.as<u32>(), b->runid.as<u32>());
      bool small_done = false;
      {  // runs of up to 32 members are sorted where they stand (run_small_sort_k)
        u32 *any_large = b->d_small + 10;
        HIP_TRY(c, hipMemsetAsync(any_large, 0, sizeof(u32), s));
        LAUNCH(run_small_sort_k, cdiv(M, 256), 256, 0, s, M, b->run_pos.as<u32>(), head, (u32)N, perm1, sorted_keys, end_bits,
               b->packed[0].as<u8>(), b->endv.as<u16>(), b->L[0], b->stride[0], ndig1, ndig, any_large);
        u32 large = 0;
        { int rc = read_u32(b, any_large, &large, 1, s); if (rc) return rc; }
        small_done = large == 0;
      }
      if (!small_done) {
      const u32 *rs = b->run_items_a.as<u32>();
      u32 *rd = b->run_items_b.as<u32>(), *ralt = b->run_items_a.as<u32>();
      auto rflip = [&]() { rs = rd; u32 *t = rd; rd = ralt; ralt = t; };
      for (int d = ndig - 1; d >= ndig1; d--) {
        radix_pass(rs, rd, M, KeyDigit{b->packed[0].as<u8>(), b->endv.as<u16>(), b->L[0], b->stride[0], d}, b->hist.as<u32>(), ws32, s);
        rflip();
      }
      int rbits = 1;
      while ((1ull << rbits) <= N && rbits < 32) rbits++;  // run ids are at most N
      for (int sh = 0; sh < rbits; sh += 8) {
        radix_pass(rs, rd, M, DigitOfArray{b->runid.as<u32>(), sh}, b->hist.as<u32>(), ws32, s);
        rflip();
      }
      LAUNCH(run_scatter_k, cdiv(M, 256), 256, 0, s, M, rs, b->run_pos.as<u32>(), perm1, sorted_keys, end_bits, b->endv.as<u16>());
      }
    }
  }
  b->perm = perm1;
  return SCALCE_OK;
}

// ---- stage 4: emit -----------------------------------------------------------------------------------
extern "C" int scalce_batch_emit(scalce_batch *b, void *stream) {
  if (!b) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  scalce_ctx *c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  StageTimer tm(b, ST_EMIT, s);
  const u64 N = b->N;
  const u32 nb1 = (u32)c->A.n_buckets + 1;
  ENSURE(b, b->bucket_first, sizeof(u64) * (nb1 + 2));
  ENSURE(b, b->bucket_off, sizeof(u64) * (nb1 + 2));
  ENSURE(b, b->scan_ws, sizeof(u64) * (scan_ws_elems(nb1) + scan_ws_elems(N + 1) + 1024));
  u64 *ws = b->scan_ws.as<u64>();
  u64 *counts = b->counts_total.as<u64>();  // reads per bucket over every piece of the batch
  if (!counts) { set_err(c, "tokenize first"); return SCALCE_ERR_ARG; }
  exclusive_scan<u64>(LoadAs<u64, u64>{counts}, nb1, StoreTo<u64>{b->bucket_first.as<u64>()}, ws, b->d_small64 + 1, s);
  exclusive_scan<u64>(BucketBytes{counts, c->d_bucket_level, b->L[0], b->sz_meta}, nb1, StoreTo<u64>{b->bucket_off.as<u64>()}, ws,
                      b->d_small64 + 2, s);
  if (b->p.use_names) {
    ENSURE(b, b->name_off, sizeof(u64) * (N + 2));
    ENSURE(b, b->outlen, N + 64);
    // the name cells are gathered through the permutation ONCE, into output order: their first byte is the length the scan
    // wants, and emit_names_sorted_k then reads them in sequence (name_outlen_k + emit_names_k gathered twice)
    // (not in lean mode: a run sized for most of HBM has no 16 bytes per read to spare, and allocating and releasing
    //  3 GB costs more than the second gather)
    const bool cells = b->namecell.p != nullptr && !b->lean;
    b->names_from_sorted_cells = cells;
    if (cells) ENSURE(b, b->cell_sorted, 16 * (N + 4));
    if (N && cells) LAUNCH(name_cells_sorted_k, cdiv(N, 256), 256, 0, s, N, b->perm, b->namecell.as<u8>(), b->cell_sorted.as<u8>(), b->outlen.as<u8>());
    else if (N) LAUNCH(name_outlen_k, cdiv(N, 256), 256, 0, s, N, b->perm, b->namelen.as<u8>(), b->outlen.as<u8>());
    exclusive_scan<u64>(NameLenSeq{b->outlen.as<u8>()}, N, StoreTo<u64>{b->name_off.as<u64>()}, ws, b->d_small64 + 3, s);
    ENSURE(b, b->bucket_name_bytes, sizeof(u64) * (nb1 + 1));
    LAUNCH(bucket_name_bytes_k, cdiv(nb1, 256), 256, 0, s, nb1, b->bucket_first.as<u64>(), counts, b->name_off.as<u64>(), b->d_small64 + 3, N,
           b->bucket_name_bytes.as<u64>());
  }
  u64 h[4] = {0, 0, 0, 0};
  { int rc = read_u64(b, b->d_small64, h, 4, s); if (rc) return rc; }
  b->out_reads_bytes[0] = h[2];
  b->out_names_bytes = b->p.use_names ? h[3] : 0;
  ENSURE(b, b->out_reads[0], h[2] + 64);
  ENSURE(b, b->out_names, b->out_names_bytes + 64);
  if (N) {
    EmitArgs a;
    a.nrec = N; a.perm = b->perm; a.bucket = b->bucket.as<u32>(); a.end = b->endv.as<u16>(); a.packed = b->packed[0].as<u8>();
    a.L = b->L[0]; a.stride = b->stride[0]; a.sz_meta = b->sz_meta; a.bucket_level = c->d_bucket_level;
    a.bucket_pattern = c->d_bucket_pattern; a.bucket_first = b->bucket_first.as<u64>(); a.bucket_off = b->bucket_off.as<u64>();
    a.counts = counts; a.out = b->out_reads[0].as<u8>();
    a.keys = b->sorted_keys; a.key_bucket_shift = b->key_bucket_shift; a.key_bucket_mask = b->key_bucket_mask;
    a.key_end_bits = b->key_end_bits;
    a.pwords = 0; a.frow = nullptr; a.cells_sorted = a.outlen = a.qs = nullptr; a.cell_off = a.qunits = 0; a.qmagic = a.rmagic = a.lmagic = 0;
    if (b->fused) {
      // One row per read: the workgroup that assembles a record's bases also moves its q' into the reordered stream -- both
      // lie in ONE row of the ingest stage's making (128 bytes = one aligned line at 100 bp), fetched whole into LDS with every
      // thread's loads in flight at once.  One random line per record instead of three (packed row + q' row for
      // gather_rows_k, each paying its own).
      ENSURE(b, b->qs(0), (size_t)b->L[0] * N + 64 + AC_INPLACE_PAD);
      a.frow = b->q[0].as<u8>(); a.stride = (int)b->qstride[0]; a.cell_off = b->row_cell_off; a.pwords = (int)b->row_pwords;
      a.packed = a.frow + b->row_cell_off;
      a.qunits = ((u32)b->L[0] + 15) / 16;
      a.qmagic = ((1ull << 32) + a.qunits - 1) / a.qunits;
      a.rmagic = ((1ull << 32) + (b->qstride[0] >> 4) - 1) / (b->qstride[0] >> 4);
      a.lmagic = ((1ull << 32) + (u32)b->L[0] - 1) / (u32)b->L[0];
      a.qs = b->qs(0).as<u8>();
      const size_t rows_lds = 256 * (size_t)b->qstride[0];
      if (rows_lds > 32 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(emit_reads_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows_lds));
      LAUNCH(emit_reads_k<true>, cdiv(N, 256), 256, rows_lds, s, a);
    } else
    LAUNCH(emit_reads_k<false>, cdiv(N, 256), 256, 0, s, a);
    if (b->p.use_names && b->names_from_sorted_cells)
      LAUNCH(emit_names_sorted_k, cdiv(N, 256), 256, 0, s, N, b->perm, b->cell_sorted.as<u8>(), b->name_in_off.as<u64>(),
             b->names_in.as<u8>(), b->name_off.as<u64>(), b->out_names.as<u8>());
    else if (b->p.use_names)
      LAUNCH(emit_names_k, cdiv(N, 256), 256, 0, s, N, b->perm, b->namecell.as<u8>(), b->name_in_off.as<u64>(),
             b->names_in.as<u8>(), b->name_off.as<u64>(), b->out_names.as<u8>());
    for (int m = 0; m < b->nm; m++) {
      const u32 w = (u32)b->L[m];
      if (m == 0 && b->fused) continue;  // (emit_reads_k<true> has done it)
      ENSURE(b, b->qs(m), (size_t)w * N + 64 + AC_INPLACE_PAD);
      LAUNCH(gather_rows_k, gather_grid(N, w), 256, 0, s, N, b->perm, b->q[m].as<u8>(), (u64)b->qstride[m], w, b->qs(m).as<u8>());
      if (b->lean) {
        // q' in input order is dead once its reordered copy exists.  Mate 1's buffer becomes mate 2's reordered stream (an
        // allocation and a release of tens of GB each cost a good part of a second), the last one is released.
        HIP_TRY(c, hipStreamSynchronize(s));
        if (m == 0 && b->nm == 2 && !b->qs(1).p && b->q[0].cap >= (size_t)b->L[1] * N + 64) {
          b->qs(1) = b->q[0];
          b->q[0] = DBuf();
        } else {
          release(b->q[m]);
        }
      }
    }
    if (b->nm == 2) {  // mate 2: bare packed reads in the same order (compress.cpp:380-383 with fR = file 4)
      const u32 w = (u32)b->szr[1];
      b->out_reads_bytes[1] = N * w;
      ENSURE(b, b->out_reads[1], N * w + 64);
      LAUNCH(gather_rows_k, gather_grid(N, w), 256, 0, s, N, b->perm, b->packed[1].as<u8>(), (u64)b->stride[1], w,
             b->out_reads[1].as<u8>());
    }
  } else if (b->nm == 2) b->out_reads_bytes[1] = 0;
  if (b->lean) {  // nothing behind this stage reads the rows, the tokens or the sort scratch
    HIP_TRY(c, hipStreamSynchronize(s));
    DBuf *dead[] = {&b->packed[0], &b->packed[1], &b->namecell, &b->names_in, &b->name_in_off, &b->name_off, &b->outlen, &b->cell_sorted,
                    &b->line_end[0], &b->line_end[1], &b->tile[0], &b->tile[1], &b->tok_bucket, &b->tok_pos, &b->tie_index,
                    &b->tie_read, &b->tie_off, &b->tie_ncand, &b->cand_bucket, &b->cand_pos, &b->choice, &b->ev_off,
                    &b->ev_bucket, &b->ev_init, &b->ev_sorted, &b->ev_tmp, &b->ev_place, &b->chosen, &b->G, &b->cand_place,
                    &b->bucket, &b->endv, &b->tokens, &b->chunk, &b->perm_a, &b->perm_b, &b->key_a, &b->key_b, &b->hist, &b->S,
                    &b->run_head, &b->run_hcount, &b->run_rank, &b->runid, &b->run_items_a, &b->run_items_b, &b->run_pos};
    for (DBuf *d : dead)
      if (d->cap >= (256u << 20)) release(*d);  // (the big ones; releasing dozens of small buffers only costs time)
    b->perm = nullptr;
    b->sorted_keys = nullptr;
    b->row_cap = 0;
  }
  return SCALCE_OK;
}

// ---- stage 5: entropy ---------------------------------------------------------------------------------
// Code one mate's symbol stream `d_sym` (nsym symbols, first symbol = start of a 10 MiB block of the run-wide
// stream) against `table` (device, 512000 x u32, already scaled).
// One coder job = one mate's symbol stream of one shard.  A launch codes the blocks of one or more jobs.
struct AcJob {
  scalce_batch *b;
  int m;
  const u8 *sym;
  u64 nsym;
  u32 nblk;
  bool general;
  u32 max_total = 0;
};
static const u64 AC_STRIDE = (u64)AC_BLOCK_SYMS + 4096;  // the reference's own output buffer is 10 MiB (arithmetic.cpp:301)
constexpr u32 AC_LOG_WORDS = 1024;        // carry notes per block coded in place (a note needs 32 ones in a row in the stream)

// table -> reciprocal fractions, buffers; one short wait for the largest context total
static int ac_prepare(AcJob &j, hipStream_t s, bool framed_output = true, bool full_stride = false, bool allow_in_place = false) {
  scalce_batch *b = j.b;
  scalce_ctx *c = b->ctx;
  const int m = j.m;
  u32 *table = b->table[m].as<u32>();
  ENSURE(b, b->ac_tab[m], sizeof(uint4) * 512000);
  ENSURE(b, b->ac_cum[m], sizeof(u32) * 6400 * 81);
  // one read-back for both: the largest context total (d_small64[12 + 4 m], low word) and the table's own coding cost
  u64 *tinfo = b->d_small64 + 12 + 4 * m;
  HIP_TRY(c, hipMemsetAsync(tinfo, 0, 3 * sizeof(u64), s));
  ENSURE(b, b->ac_tab8[m], sizeof(u64) * (6400 * 81 + 2));
  LAUNCH(ac_table_k, cdiv(6400, 64), 64, 0, s, table, b->ac_tab[m].as<uint4>(), b->ac_cum[m].as<u32>(), reinterpret_cast<u32 *>(tinfo), b->ac_tab8[m].as<u64>(),
         reinterpret_cast<unsigned long long *>(tinfo + 1));
  u64 th[3] = {0, 0, 0};
  { int rc = read_u64(b, tinfo, th, 3, s); if (rc) return rc; }
  const u32 max_total = (u32)th[0];
  // above 2^30 a symbol's interval can collapse in the reference's 32-bit coder; only the general step
  // follows it there bit for bit
  j.general = max_total > (1u << 30) || getenv("SCALCE_AC_GENERAL") != nullptr;
  j.max_total = max_total;
  j.nblk = (u32)cdiv(j.nsym, AC_BLOCK_SYMS);
  {
    u64 stride = AC_STRIDE;
    const char *scale_env = getenv("SCALCE_AC_STRIDE_SCALE");  // test hook: "0" = the reference's full stride, else a factor (too small on purpose)
    const bool full = scale_env && atof(scale_env) == 0.0;
    if (!full && !full_stride && th[2]) {
      double bytes_per_symbol = (double)th[1] / 256.0 / 8.0 / (double)th[2];
      if (scale_env) bytes_per_symbol *= atof(scale_env);
      const u64 est = (u64)((double)AC_BLOCK_SYMS * bytes_per_symbol * 1.08) + 65536;
      stride = std::min<u64>(AC_STRIDE, (est + 15) & ~15ull);
    }
    b->ac_stride[m] = stride;
  }
  // in place: only the batch's own reordered stream, whole (the kernels of a grouped launch all know how; ac_encode_k does not)
  b->in_place_now[m] = allow_in_place && b->code_in_place && !b->in_place_suspended && !b->qs_in_ws && j.nblk &&
                       j.sym == b->qs(m).as<u8>() && b->qs(m).cap >= j.nsym + 64 + AC_INPLACE_PAD;
  if (b->in_place_now[m]) {
    b->ac_stride[m] = AC_BLOCK_SYMS;
    b->ac_base[m] = b->qs(m).as<u8>();
    ENSURE(b, b->ac_log[m], sizeof(u32) * AC_LOG_WORDS * (size_t)j.nblk + 64);
  } else {
    ENSURE(b, b->ac_blocks[m], (size_t)j.nblk * b->ac_stride[m] + 4096 + 64);  // (the frame kernels read a few words past a block's bytes)
    b->ac_base[m] = b->ac_blocks[m].as<u8>();
  }
  ENSURE(b, b->ac_sizes[m], sizeof(u32) * (j.nblk + 2));
  ENSURE(b, b->ac_off[m], sizeof(u64) * (j.nblk + 2));
  // the framed stream is sized for the worst case (every block at its cap): no size has to come back from the
  // device before the frame kernel can be enqueued
  if (framed_output && !b->frame_on_demand) ENSURE(b, b->out_qual[m], (size_t)j.nblk * (b->ac_stride[m] + 4) + 64);
  ENSURE(b, b->ac_scan, sizeof(u64) * (scan_ws_elems(j.nblk ? j.nblk : 1) + 64));  // (the batch's own: framing runs at collect time,
  if (!j.nblk) b->out_qual_bytes[m] = 0;                                            //  beside another batch's front stages)
  return SCALCE_OK;
}

// Blocks per workgroup of ac_encode_lanes_k.  A lane per block would be 64; the default is fewer: all table rows of a
// workgroup's blocks go through ONE CU's vector memory pipeline (1024 scattered 16-byte reads per 0.75 us round at 64), and
// beside another shard's streaming front stages -- when an L2 miss takes three times as long -- that pipeline, not the coder,
// set the pace of a launch: 908 ms beside the ingest stage and 1299 ms beside the order stage at 64 blocks per workgroup
// against 575 / 694 ms at 32 and 560 / 559 ms at 16 (543 ms alone; tools/coder_beside.py).  Fewer blocks per workgroup
// are more CUs held per launch, CUs the front stages of the next shards do not get: in round 3's mix 48 was the best
// trade (ms per shard at 32 / 40 / 48 / 56 / 64 blocks: 107.0 / 106.6 / 102.4 / 108.0 / 115.5); since
// round 4 the default is 40 (below).
static u32 ac_lanes_used() {
  const char *e = getenv("SCALCE_AC_LANES_USED");
  // round 4 (twelve slots, block buffers of their own): 79.9 / 80.0 / 81.1 / 81.7 / 82.2 / 85.6 ms per shard at 32 / 36 / 40 / 44 / 48 / 56,
  // 40 the default.  Round 5 (fifteen slots, coded in place, six shards per launch on two streams; tools/r5_sweep.sh): 79.3 / 79.6 /
  // 74.0 / 74.6 / 75.2 / 77.9 / 81.8 at 24 / 28 / 32 / 36 / 40 / 48 / 56 -- a launch takes 0.47 s at 32 against 0.50 at 40, and with
  // slots to spare the pipeline follows the launch's latency: 32.
  const int v = e ? atoi(e) : 32;
  return (u32)(v < 1 || v > 64 ? 32 : v);
}

// ONE launch over the blocks of all jobs.  blocks_per_wg: 1 = ac_encode_k (one job only), 4 / 8 = ac_encode_rows_k,
// 64 = ac_encode_lanes_k (one block per lane).
// `ps` = the stream the tables were prepared on: the block descriptors are uploaded there (never behind a coder that
// is still running on `s`), and `s` is made to wait for it.
static int ac_launch(AcJob *jobs, int njobs, int blocks_per_wg, hipStream_t s, hipStream_t ps) {
  scalce_batch *lead = jobs[0].b;
  scalce_ctx *c = lead->ctx;
  u32 total = 0;
  bool general = false;
  for (int i = 0; i < njobs; i++) {
    total += jobs[i].nblk;
    general |= jobs[i].general;
    // the plain path's exit test (see ac_encode_k) needs every symbol to keep an interval of two values or more
    if (jobs[i].max_total > (1u << 29)) general = true;
  }
  if (!total) return SCALCE_OK;
  AcEncArgs a;
  memset(&a, 0, sizeof a);
  a.slow_threshold = 32;
  a.chain_prio = 3u;
  a.helper_prio = 0u;
  a.test_poison = getenv("SCALCE_AC_TEST_POISON") ? (u32)atoi(getenv("SCALCE_AC_TEST_POISON")) : 0u;  // test hook
  a.inplace_shift = getenv("SCALCE_AC_INPLACE_TEST") ? 2u : 0u;  // test hook: a block coded in place catches up with its input
  a.simd_load = c->d_simd_load;
  if (const char *e = getenv("SCALCE_AC_SLOW_THRESHOLD")) a.slow_threshold = (u32)atoi(e);  // test hook
  auto join = [&]() -> int {  // `s` continues behind everything enqueued on `ps` so far
    if (ps == s) return SCALCE_OK;
    hipEvent_t ev = lead->ev_group;
    if (!ev) { HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming)); lead->ev_group = ev; }
    HIP_TRY(c, hipEventRecord(ev, ps));
    HIP_TRY(c, hipStreamWaitEvent(s, ev, 0));
    return SCALCE_OK;
  };
  hipEvent_t ke0 = nullptr, ke1 = nullptr;
  if (lead->ktiming) {
    if (lead->kev_used == lead->kev.size()) {
      hipEvent_t x, y;
      HIP_TRY(c, hipEventCreate(&x));
      HIP_TRY(c, hipEventCreate(&y));
      lead->kev.emplace_back(x, y);
    }
    ke0 = lead->kev[lead->kev_used].first; ke1 = lead->kev[lead->kev_used].second;
    lead->kev_used++;
  }
  if (blocks_per_wg == 1) {
    if (njobs != 1) { set_err(c, "internal: ac_encode_k takes one job"); return SCALCE_ERR_ARG; }
    scalce_batch *b = jobs[0].b;
    const int m = jobs[0].m;
    a.sym = jobs[0].sym; a.nsym = jobs[0].nsym; a.tab = b->ac_tab[m].as<uint4>(); a.out = b->ac_base[m];
    a.out_stride = b->ac_stride[m]; a.out_cap = (u32)b->ac_stride[m]; a.out_size = b->ac_sizes[m].as<u32>(); a.err = b->d_err;
    if (getenv("SCALCE_AC_PROF")) { HIP_TRY(c, hipMalloc(&a.prof, sizeof(u64) * 3 * total)); }
    { int rc = join(); if (rc) return rc; }
    if (ke0) hipEventRecord(ke0, s);
    if (general) LAUNCH(ac_encode_k<true>, total, 128, 0, s, a);
    else LAUNCH(ac_encode_k<false>, total, 128, 0, s, a);
    if (ke1) hipEventRecord(ke1, s);
    if (a.prof) {  // profiling only: where do the chain wave's cycles go (shader clock)
      std::vector<u64> h(3 * (size_t)total);
      HIP_TRY(c, hipMemcpy(h.data(), a.prof, sizeof(u64) * h.size(), hipMemcpyDeviceToHost));
      double sys = 0, tot = 0, rounds = 0;
      for (u32 i = 0; i < total; i++) { sys += h[3 * i]; tot += h[3 * i + 1]; rounds += h[3 * i + 2]; }
      fprintf(stderr, "ac prof: %u blocks, per plain round: %.1f cycles in the 64 steps, %.1f cycles in all (%.0f plain rounds per block)\n",
              total, sys / rounds, tot / rounds, rounds / total);
      hipFree(a.prof);
    }
  } else {
    // block descriptors: the launch may hold blocks of several shards, each with its own table and output
    if (total > lead->ac_desc_cap) {
      if (lead->ac_desc_host) hipHostFree(lead->ac_desc_host);
      lead->ac_desc_host = nullptr;
      lead->ac_desc_cap = 0;
      // (room for any launch this shard is likely to lead: an allocation synchronises the whole device -- a shard that led a
      //  launch of one shard and then one of three waited there for every coder that was running)
      const u32 cap = std::max<u32>(total + total / 2, 8192u);
      HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&lead->ac_desc_host), sizeof(AcBlockDesc) * (size_t)cap, hipHostMallocDefault));
      lead->ac_desc_cap = cap;
    }
    // one block per lane only follows the reference while no interval can invert (kernels_acl.hpp)
    if (blocks_per_wg == 64 && general) blocks_per_wg = 8;
    const bool lanes = blocks_per_wg == 64;  // gathers from the compact table
    AcBlockDesc *d = lead->ac_desc_host;
    u32 nd = 0;
    for (int i = 0; i < njobs; i++) {
      scalce_batch *b = jobs[i].b;
      const int m = jobs[i].m;
      for (u32 k = 0; k < jobs[i].nblk; k++) {
        AcBlockDesc x;
        const u64 off = (u64)k * AC_BLOCK_SYMS;
        x.sym = jobs[i].sym + off;
        x.tab = lanes ? reinterpret_cast<const uint4 *>(b->ac_tab8[m].as<u64>()) : b->ac_tab[m].as<uint4>();
        x.dst = reinterpret_cast<u32 *>(b->ac_base[m] + (u64)k * b->ac_stride[m]);
        x.cap = (u32)b->ac_stride[m];
        x.flags = 0; x.log = nullptr; x.log_cap = 0; x.pad_ = 0;
        if (b->in_place_now[m]) {   // the block's own symbols are its buffer; the last block of a stream may run into the padding
          const u64 left = jobs[i].nsym - (u64)k * AC_BLOCK_SYMS;
          if (left < AC_BLOCK_SYMS) x.cap = (u32)std::min<u64>(AC_BLOCK_SYMS, (left + AC_INPLACE_PAD) & ~3ull);
          x.flags = AC_BLOCK_IN_PLACE;
          x.log = b->ac_log[m].as<u32>() + (size_t)k * AC_LOG_WORDS;
          x.log_cap = AC_LOG_WORDS;
        }
        x.out_size = b->ac_sizes[m].as<u32>() + k;
        x.err = b->d_err;
        x.n = (u32)std::min<u64>(AC_BLOCK_SYMS, jobs[i].nsym - off);
        x.index = k;
        d[nd++] = x;
      }
    }
    ENSURE(lead, lead->ac_desc, sizeof(AcBlockDesc) * (size_t)lead->ac_desc_cap);
    HIP_TRY(c, hipMemcpyAsync(lead->ac_desc.p, d, sizeof(AcBlockDesc) * total, hipMemcpyHostToDevice, ps));
    a.desc = lead->ac_desc.as<AcBlockDesc>();
    a.nblocks = total;
    a.out_cap = (u32)AC_STRIDE;
    { int rc = join(); if (rc) return rc; }
    if (ke0) hipEventRecord(ke0, s);
    const u32 nwg = cdiv(total, blocks_per_wg == 64 ? ac_lanes_used() : (u32)blocks_per_wg);
    if (getenv("SCALCE_AC_PROF")) { HIP_TRY(c, hipMalloc(&a.prof, sizeof(u64) * 5 * (nwg + 2))); HIP_TRY(c, hipMemset(a.prof, 0, sizeof(u64) * 5 * (nwg + 2))); }
    if (blocks_per_wg == 64) {
      a.lanes_used = ac_lanes_used();
      // (two sets of four waves per CU -- a chain or sink sharing its SIMD with a light wave of the other set -- cost a third fewer
      //  CU-seconds and 35 % more latency per launch: 98 against 75 ms per shard with fifteen slots, round 5; waves on shared CUs at
      //  raised priority: +5 %, round 3.  Both removed.)
      if (a.lanes_used <= 48) LAUNCH((ac_encode_lanes_k<true, 1, 48, 5>), cdiv(total, a.lanes_used), 256, 0, s, a);  // (rows of 48: LDS for a longer staging ring)
      else LAUNCH((ac_encode_lanes_k<true, 1, 64, 5>), cdiv(total, a.lanes_used), 256, 0, s, a);
    } else if (blocks_per_wg == 8) {
      if (general) LAUNCH((ac_encode_rows_k<true, 8>), cdiv(total, 8), 320, 0, s, a);
      else LAUNCH((ac_encode_rows_k<false, 8>), cdiv(total, 8), 320, 0, s, a);
    } else {
      if (general) LAUNCH((ac_encode_rows_k<true, 16>), cdiv(total, 4), 192, 0, s, a);
      else LAUNCH((ac_encode_rows_k<false, 16>), cdiv(total, 4), 192, 0, s, a);
    }
    if (ke1) hipEventRecord(ke1, s);
    if (a.prof) {  // profiling only: read back when the lead shard is collected (the launch keeps running beside others)
      lead->prof_ptr = a.prof;
      lead->prof_n = nwg;
      lead->prof_lanes = blocks_per_wg == 64;
    }
  }
  for (int i = 0; i < njobs; i++) {
    lead->k_in_bytes += jobs[i].nsym;
    jobs[i].b->ac_last_sym[jobs[i].m] = jobs[i].sym;
    jobs[i].b->ac_last_nsym[jobs[i].m] = jobs[i].nsym;
  }
  return SCALCE_OK;
}

// framing: sizes -> offsets -> [u32 size][bytes] per block, all enqueued; the total is read back by entropy_collect
static bool frames_at_collect() { return true; }  // (behind the coder on its own stream: measured slower)
static int ac_frame(AcJob &j, hipStream_t s) {
  scalce_batch *b = j.b;
  const int m = j.m;
  b->frame_virtual[m] = 0;
  if (!j.nblk) return SCALCE_OK;
  exclusive_scan<u64>(AcFrameLen{b->ac_sizes[m].as<u32>()}, j.nblk, StoreTo<u64>{b->ac_off[m].as<u64>()}, b->ac_scan.as<u64>(),
                      b->d_small64 + 8 + m, s);
  if (b->frame_on_demand) {  // the layout is all there is for now
    b->frame_virtual[m] = j.nblk;
    b->ent_pending[m] = j.nblk;
    return SCALCE_OK;
  }
  if (b->out_qual[m].cap < (size_t)j.nblk * (b->ac_stride[m] + 4) + 64) {
    // the framed stream was not sized for the worst case (a grouped launch: twelve and more shards in flight, and 5 GB each
    // of a capacity that is little more than half used is a shard less in flight): the size comes back first -- the coder has
    // finished, this is a wait of microseconds -- and the buffer grows when a shard codes worse than any before it
    u64 total = 0;
    { int rc = read_u64(b, b->d_small64 + 8 + m, &total, 1, s); if (rc) return rc; }
    if (b->out_qual[m].cap < total + 64) ENSURE(b, b->out_qual[m], (size_t)(total + total / 16) + (32u << 20));
  }
  LAUNCH(ac_frame_k, dim3(cdiv(b->ac_stride[m], 16 * 256), j.nblk), 256, 0, s, b->ac_base[m], b->ac_stride[m],
         b->ac_sizes[m].as<u32>(), b->ac_off[m].as<u64>(), b->out_qual[m].as<u8>());
  b->ent_pending[m] = j.nblk;
  return SCALCE_OK;
}

static int ac_blocks_per_wg() {
  // one block per workgroup (lowest latency of a block) or four (0.57 x the SIMD time per block, 1.2 x the latency)
  const char *bpw = getenv("SCALCE_AC_BLOCKS_PER_WG");
  const int v = bpw ? atoi(bpw) : 1;
  return (v == 4 || v == 8 || v == 64) ? v : 1;
}

// Code one mate's symbol stream `d_sym` (nsym symbols, first symbol = start of a 10 MiB block of the run-wide
// stream) against `table` (device, 512000 x u32, already scaled).
static int encode_stream(scalce_batch *b, int m, const u8 *d_sym, u64 nsym, hipStream_t s) {
  AcJob j{b, m, d_sym, nsym, 0, false};
  int rc = ac_prepare(j, s);
  if (rc) return rc;
  if ((rc = ac_launch(&j, 1, ac_blocks_per_wg(), s, s))) return rc;
  return ac_frame(j, s);
}

// A block outgrew the stride its shard's table suggested (ac_prepare): the shard's streams are coded again with the
// reference's own 10 MiB per block, now, on the collecting stream.  Rare by construction; the same bytes one launch later.
static int entropy_recode_full(scalce_batch *b, hipStream_t s) {
  std::vector<AcJob> jobs;
  for (int m = 0; m < b->nm; m++) {
    if (!b->ac_last_sym[m] || !b->ac_last_nsym[m]) continue;
    AcJob j{b, m, b->ac_last_sym[m], b->ac_last_nsym[m], 0, false};
    int rc = ac_prepare(j, s, /*framed_output=*/false, /*full_stride=*/true);
    if (rc) return rc;
    jobs.push_back(j);
  }
  if (jobs.empty()) return SCALCE_OK;
  u32 total = 0;
  for (auto &j : jobs) total += j.nblk;
  int rc = ac_launch(jobs.data(), (int)jobs.size(), total <= 1024 ? 4 : 8, s, s);
  if (rc) return rc;
  for (auto &j : jobs) { b->frame_deferred[j.m] = j.nblk; b->ent_pending[j.m] = 0; }
  return SCALCE_OK;
}

// A block coded in place caught up with its own input (kernels_acl.hpp, writer wave; kernels_ac.hpp, helper waves): the symbols
// it had consumed are under its output, nothing can be coded again from them.  The shard is run again from its TEXT -- which a
// caller that turns coding in place on keeps where it was until the shard is collected -- with block buffers of its own.
// Never seen on quality strings (their code is shorter than their symbols by a third and more from the first round on);
// SCALCE_AC_INPLACE_TEST=1 makes the kernels' bound so tight that it happens (tests).
static int entropy_rerun_from_text(scalce_batch *b, hipStream_t s) {
  scalce_ctx *c = b->ctx;
  if (b->appending || b->base != 0 || !b->piece_text[0] || (b->nm == 2 && !b->piece_text[1])) {
    set_err(c, "a block coded in place outgrew its input and the shard's text is not at hand to run it again (scalce_batch_set_code_in_place)");
    return SCALCE_ERR_CAPACITY;
  }
  static const bool dbg = getenv("SCALCE_DEBUG_ALLOC") != nullptr;
  if (dbg) fprintf(stderr, "scalce: batch %p: a block coded in place caught up with its input: the shard is run again from its text\n", (void *)b);
  const u8 *t1 = b->piece_text[0], *t2 = b->nm == 2 ? b->piece_text[1] : nullptr;
  const u64 n1 = b->text_bytes[0], n2 = b->nm == 2 ? b->text_bytes[1] : 0;
  for (int m = 0; m < 2; m++) { b->frame_deferred[m] = 0; b->ent_pending[m] = 0; b->in_place_now[m] = false; }
  b->in_place_suspended = true;
  b->reruns++;
  int rc = scalce_batch_front(b, t1, n1, t2, n2, s);
  if (!rc) rc = scalce_batch_entropy(b, nullptr, s);
  b->in_place_suspended = false;
  return rc;
}

// second half of the entropy stage: wait for the coder and read the size of the framed stream(s)
static int entropy_collect(scalce_batch *b, hipStream_t s) {
  {
    bool open = false, tight = false, in_place = false;
    for (int m = 0; m < b->nm; m++) {
      open |= b->frame_deferred[m] != 0 || b->ent_pending[m] != 0;
      tight |= b->ac_stride[m] != 0 && b->ac_stride[m] < AC_STRIDE;
      in_place |= b->in_place_now[m];
    }
    if (open && tight) {
      DevErr e;
      HIP_TRY(b->ctx, hipMemcpyAsync(&e, b->d_err, sizeof e, hipMemcpyDeviceToHost, s));
      HIP_TRY(b->ctx, hipStreamSynchronize(s));
      if (e.code == E_ACOVERFLOW) {
        if (getenv("SCALCE_DEBUG_ALLOC"))
          fprintf(stderr, "scalce: batch %p: block %llu outgrew its %llu-byte buffer (it needed %u): coding the shard again at the full stride\n",
                  (void *)b, (unsigned long long)e.where, (unsigned long long)b->ac_stride[0], e.aux);
        HIP_TRY(b->ctx, hipMemsetAsync(b->d_err, 0, sizeof(DevErr), s));
        if (in_place) return entropy_rerun_from_text(b, s);   // (collects by itself)
        int rc = entropy_recode_full(b, s);
        if (rc) return rc;
      }
    }
  }
  if (b->prof_ptr) {  // profiling only: share of the chain waves' time spent waiting at the barrier
    std::vector<u64> h(5 * (size_t)b->prof_n);
    HIP_TRY(b->ctx, hipMemcpy(h.data(), b->prof_ptr, sizeof(u64) * h.size(), hipMemcpyDeviceToHost));
    double wait = 0, tot = 0, hwait = 0, htot = 0;
    for (u32 i = 0; i < b->prof_n; i++) { wait += h[5 * i]; tot += h[5 * i + 1]; hwait += h[5 * i + 3]; htot += h[5 * i + 4]; }
    fprintf(stderr, "ac prof (rows): %u workgroups, chain waves waited at the barrier %.1f %% of their time (%.0f of %.0f Mcycles each), "
            "the first helper wave %.1f %%\n",
            b->prof_n, 100.0 * wait / tot, wait / b->prof_n / 1e6, tot / b->prof_n / 1e6, 100.0 * hwait / (htot > 0 ? htot : 1));
    if (b->prof_lanes) {
      fprintf(stderr, "ac prof (lanes): SIMD of chain / gather / sink / writer per workgroup:");
      for (u32 i = 0; i < b->prof_n && i < 24; i++) fprintf(stderr, " %llu%llu%llu%llu", h[5 * i + 2] & 15, (h[5 * i + 2] >> 4) & 15, (h[5 * i + 2] >> 8) & 15, (h[5 * i + 2] >> 12) & 15);
      fprintf(stderr, "\n");
    }
    if (b->prof_lanes)
      fprintf(stderr, "ac prof (lanes): of the chain wave's %.0f Mcycles the gather wave waited at the barrier %.1f %%, the sink wave %.1f %%\n",
              tot / b->prof_n / 1e6, 100.0 * hwait / tot, 100.0 * htot / tot);
    hipFree(b->prof_ptr);
    b->prof_ptr = nullptr;
  }
  for (int m = 0; m < b->nm; m++) {
    if (b->frame_deferred[m]) {
      AcJob j{b, m, nullptr, 0, b->frame_deferred[m], false};
      b->frame_deferred[m] = 0;
      int rc = ac_frame(j, s);
      if (rc) return rc;
    }
    if (!b->ent_pending[m]) continue;
    u64 total = 0;
    if (b->frame_virtual[m]) {  // (one wait for both: the layout and the total)
      b->frame_off_host[m].resize(b->frame_virtual[m]);
      HIP_TRY(b->ctx, hipMemcpyAsync(b->frame_off_host[m].data(), b->ac_off[m].p, sizeof(u64) * b->frame_virtual[m], hipMemcpyDeviceToHost, s));
    }
    { int rc = read_u64(b, b->d_small64 + 8 + m, &total, 1, s); if (rc) return rc; }
    b->out_qual_bytes[m] = total;
    b->k_out_bytes += total - 4ull * b->ent_pending[m];
    b->ent_pending[m] = 0;
  }
  return SCALCE_OK;
}

// table of one mate: the shard's own statistics, scaled (compress.cpp:297-303), or the caller's run-wide table
static int ac_table_for(scalce_batch *b, int m, const uint32_t *d_table_override, u64 nsym, hipStream_t s) {
  scalce_ctx *c = b->ctx;
  u32 *table = b->table[m].as<u32>();
  if (d_table_override) {
    HIP_TRY(c, hipMemcpyAsync(table, d_table_override + (size_t)m * 512000, sizeof(u32) * 512000, hipMemcpyDeviceToDevice, s));
  } else {
    const u32 factor = 1 + (u32)(nsym / 0xFFFFFFFFull);  // compress.cpp:297-303
    LAUNCH(ac_scale_k, cdiv(512000, 256), 256, 0, s, b->freq4[m].as<u64>(), factor, table);
  }
  return SCALCE_OK;
}

// Runs with more blocks than a launch should hold (200 M x 150 bp paired: 5 724): the streams are coded window by window --
// up to AC_WINDOW_BLOCKS blocks per launch over both mates -- into block buffers sized for one window, and every window
// is framed straight behind the previous one.  Sized for the worst case as the one-launch path does, the buffers of
// such a run would take 2 x 60 GB per mate.
constexpr u32 AC_WINDOW_BLOCKS = 2048;  // 256 workgroups of eight
static int entropy_windowed(scalce_batch *b, const uint32_t *d_table_override, hipStream_t s) {
  scalce_ctx *c = b->ctx;
  const u64 N = b->N;
  u32 W = AC_WINDOW_BLOCKS / (u32)b->nm;
  if (const char *e = getenv("SCALCE_AC_WINDOW_BLOCKS")) W = (u32)std::max(1, atoi(e));  // test hook: blocks per mate and window
  u64 used[2] = {0, 0}, nsym[2] = {0, 0};
  u32 nblk[2] = {0, 0}, most = 0;
  for (int m = 0; m < b->nm; m++) {
    nsym[m] = N * (u64)b->L[m];
    nblk[m] = cdiv(nsym[m], AC_BLOCK_SYMS);
    most = nblk[m] > most ? nblk[m] : most;
    b->frame_virtual[m] = 0;   // this path writes the framed stream itself: a layout left by an earlier shard is void
    b->frame_off_host[m].clear();
    int rc = ac_table_for(b, m, d_table_override, nsym[m], s);
    if (rc) return rc;
    b->out_qual_bytes[m] = 0;
  }
  bool first = true;
  for (u32 w0 = 0; w0 < most; w0 += W) {
    AcJob jobs[2];
    int nj = 0;
    for (int m = 0; m < b->nm; m++) {
      if (w0 >= nblk[m]) continue;
      const u64 off = (u64)w0 * AC_BLOCK_SYMS;
      const u64 n = std::min<u64>(nsym[m] - off, (u64)W * AC_BLOCK_SYMS);
      jobs[nj] = AcJob{b, m, b->qs(m).as<u8>() + off, n, 0, false};
      int rc = ac_prepare(jobs[nj], s, /*framed_output=*/false, /*full_stride=*/true);  // (windows are collected as they go: no second try)
      if (rc) return rc;
      nj++;
    }
    if (!nj) break;
    int rc = ac_launch(jobs, nj, 8, s, s);
    if (rc) return rc;
    for (int i = 0; i < nj; i++) {
      const int m = jobs[i].m;
      exclusive_scan<u64>(AcFrameLen{b->ac_sizes[m].as<u32>()}, jobs[i].nblk, StoreTo<u64>{b->ac_off[m].as<u64>()}, b->ac_scan.as<u64>(),
                          b->d_small64 + 8 + m, s);
      u64 total = 0;
      if ((rc = read_u64(b, b->d_small64 + 8 + m, &total, 1, s))) return rc;
      if (first) {  // size the framed stream from the first window's ratio; it grows if a later window codes worse
        const u64 est = (u64)((double)total / (double)jobs[i].nsym * 1.03 * (double)nsym[m]) + (64u << 20);
        ENSURE(b, b->out_qual[m], est);
      }
      if ((rc = ensure_keep(b, b->out_qual[m], used[m] + total + 64, used[m], s))) return rc;
      LAUNCH(ac_frame_k, dim3(cdiv(b->ac_stride[m], 16 * 256), jobs[i].nblk), 256, 0, s, b->ac_base[m], b->ac_stride[m],
             b->ac_sizes[m].as<u32>(), b->ac_off[m].as<u64>(), b->out_qual[m].as<u8>() + used[m]);
      used[m] += total;
      b->k_out_bytes += total - 4ull * jobs[i].nblk;
    }
    first = false;
  }
  HIP_TRY(c, hipStreamSynchronize(s));
  for (int m = 0; m < b->nm; m++) { b->out_qual_bytes[m] = used[m]; b->ent_pending[m] = 0; }
  return SCALCE_OK;
}

extern "C" int scalce_batch_entropy_begin(scalce_batch *b, const uint32_t *d_table_override, void *stream) {
  if (!b) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  scalce_ctx *c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  const u64 N = b->N;
  if (!b->p.no_ac) {
    u64 blocks = 0;
    for (int m = 0; m < b->nm; m++) blocks += cdiv(N * (u64)b->L[m], AC_BLOCK_SYMS);
    if (blocks > AC_WINDOW_BLOCKS || getenv("SCALCE_AC_WINDOW_BLOCKS")) return entropy_windowed(b, d_table_override, s);  // (the variable: a test hook)
  }
  if (b->nm == 2 && !b->p.no_ac && ac_blocks_per_wg() == 1) {
    // paired reads: both mates' streams in ONE launch (several blocks per chain wave) instead of two launches of the
    // one-block kernel behind each other -- the same chip, half the time
    AcJob jobs[2];
    u32 total = 0;
    for (int m = 0; m < 2; m++) {
      const u64 nsym = N * (u64)b->L[m];
      int rc = ac_table_for(b, m, d_table_override, nsym, s);
      if (rc) return rc;
      jobs[m] = AcJob{b, m, b->qs(m).as<u8>(), nsym, 0, false};
      if ((rc = ac_prepare(jobs[m], s))) return rc;
      total += jobs[m].nblk;
    }
    int rc = ac_launch(jobs, 2, total <= 4 * 256 ? 4 : 8, s, s);
    if (rc) return rc;
    for (int m = 0; m < 2; m++)
      if ((rc = ac_frame(jobs[m], s))) return rc;
    return SCALCE_OK;
  }
  for (int m = 0; m < b->nm; m++) {
    const u64 nsym = N * (u64)b->L[m];
    if (b->p.no_ac) {  // -A: raw q' bytes (compress.cpp:389-390)
      b->out_qual_bytes[m] = nsym;
      continue;
    }
    int rc = ac_table_for(b, m, d_table_override, nsym, s);
    if (rc) return rc;
    if ((rc = encode_stream(b, m, b->qs(m).as<u8>(), nsym, s))) return rc;
  }
  return SCALCE_OK;
}

// Several shards, ONE coder launch (four blocks per workgroup): a shard of 477 blocks fills 120 workgroups, so
// the blocks of two shards fit the 256 CUs one workgroup each -- every chain wave gets a SIMD of its own by
// construction, which two separate launches cannot guarantee (the dispatcher places waves without knowing their
// role; where two chains meet the younger one starves).  Tables are prepared on `prep_stream` (the caller's front
// stream: the host waits there for each table's largest context total, never behind a running coder); coder and
// framing are enqueued on `stream` behind that.  Shards that called scalce_batch_entropy_stream_prepare code the
// stream they were given, the others their own reordered stream.
extern "C" int scalce_batch_entropy_begin_group(scalce_batch **bs, int n, void *prep_stream, void *stream) {
  return scalce_batch_entropy_begin_group_last(bs, n, prep_stream, stream, 0);
}
// last != 0: nothing will be queued behind this launch (the end of a run): what counts is how soon it is done, not how few
// CUs it holds -- eight blocks per chain wave (0.36-0.47 s for up to 2048 blocks) instead of one block per lane (0.56-0.65 s)
extern "C" int scalce_batch_entropy_begin_group_last(scalce_batch **bs, int n, void *prep_stream, void *stream, int last) {
  if (!bs || n <= 0 || n > 16) return SCALCE_ERR_ARG;
  for (int i = 0; i < n; i++) if (!bs[i] || bs[i]->ctx != bs[0]->ctx) return SCALCE_ERR_ARG;
  hipStream_t ps = (hipStream_t)prep_stream, s = (hipStream_t)stream;
  scalce_ctx *c = bs[0]->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<AcJob> jobs;
  for (int i = 0; i < n; i++) {
    scalce_batch *b = bs[i];
    for (int m = 0; m < b->nm; m++) {
      const u64 own = b->N * (u64)b->L[m];
      if (b->p.no_ac) { b->out_qual_bytes[m] = own; continue; }
      AcJob j{b, m, b->ent_external[m] ? b->ent_sym[m] : b->qs(m).as<u8>(), b->ent_external[m] ? b->ent_nsym[m] : own, 0, false};
      if (!b->ent_external[m]) { int rc = ac_table_for(b, m, nullptr, own, ps); if (rc) return rc; }
      b->ent_external[m] = false;
      int rc = ac_prepare(j, ps, /*framed_output=*/!frames_at_collect(), /*full_stride=*/false, /*allow_in_place=*/true);
      if (rc) return rc;
      jobs.push_back(j);
    }
  }
  if (jobs.empty()) return SCALCE_OK;
  // blocks per workgroup: as few as keep the launch at one workgroup per CU (256 CUs), so that every coder wave has
  // a SIMD to itself -- four for up to two 50 M-read shards, eight beyond
  u32 total = 0;
  for (auto &j : jobs) total += j.nblk;
  // one shard (477 blocks at 50 M x 100): four blocks per chain wave, the lowest latency that still leaves every chain wave a
  // SIMD of its own; from two shards on one block per LANE -- the launch then takes ~0.56 s whatever its size, but on a
  // sixth of the SIMD time per block, and the front stages of the next shards keep the chip
  // last == 2: a small launch at the START of a run (more shards are on their way: the CUs belong to their front stages):
  // one block per lane whatever the size
  // last == 3: one of the last launches of a run, with a front stage or two still to come: eight blocks per chain wave (60 CUs
  // per 50 M-read shard for ~0.4 s) -- sooner done than one block per lane, and not the whole chip
  int bpw = (total < 900 && last != 2 && last != 3) ? 4 : 64;
  if ((last == 1 && total >= 900 && total <= 2048) || (last == 3 && total <= 1024)) bpw = 8;
  if (ac_blocks_per_wg() != 1) bpw = ac_blocks_per_wg();

  int rc = ac_launch(jobs.data(), (int)jobs.size(), bpw, s, ps);
  if (rc) return rc;
  // The framing ([u32 size][bytes] per block: a scan of the sizes + one copy kernel, ~1.5 ms per 50 M-read shard on an
  // idle chip) is left to entropy_collect, i.e. to the stream the caller collects on.  Behind the coder on its own stream
  // (SCALCE_AC_FRAME_BEHIND_CODER=1) it was measured slower with one coder stream (it lengthens the launch the pipeline
  // waits for) and with three (95.6 against 93.0 ms per shard).
  const bool at_collect = frames_at_collect();
  for (auto &j : jobs) {
    if (at_collect) { j.b->frame_deferred[j.m] = j.nblk; continue; }
    rc = ac_frame(j, s);
    if (rc) return rc;
  }
  return SCALCE_OK;
}

// Sharded runs, grouped launch: remember the caller-assembled range of the run-wide stream and its table; the next
// scalce_batch_entropy_begin_group codes it.
extern "C" int scalce_batch_entropy_stream_prepare(scalce_batch *b, int mate, const uint32_t *d_table, const uint8_t *d_symbols,
                                                   uint64_t nsym, void *stream) {
  if (!b || mate < 0 || mate >= b->nm || !d_table || (nsym && !d_symbols) || b->p.no_ac) return SCALCE_ERR_ARG;
  scalce_ctx *c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(b->table[mate].p, d_table, sizeof(u32) * 512000, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  b->ent_sym[mate] = d_symbols;
  b->ent_nsym[mate] = nsym;
  b->ent_external[mate] = true;
  return SCALCE_OK;
}

extern "C" int scalce_batch_entropy_end(scalce_batch *b, void *stream) {
  if (!b) return SCALCE_ERR_ARG;
  HIP_TRY(b->ctx, hipSetDevice(b->ctx->device));
  return entropy_collect(b, (hipStream_t)stream);
}

extern "C" int scalce_batch_entropy(scalce_batch *b, const uint32_t *d_table_override, void *stream) {
  if (!b) return SCALCE_ERR_ARG;
  StageTimer tm(b, ST_ENTROPY, (hipStream_t)stream);
  int rc = scalce_batch_entropy_begin(b, d_table_override, stream);
  if (rc) return rc;
  return entropy_collect(b, (hipStream_t)stream);
}

// Sharded runs: code `nsym` symbols of mate `mate` that the caller assembled on the device (a range of the
// run-wide reordered stream that starts on a 10 MiB block boundary) against the run-wide table.
extern "C" int scalce_batch_entropy_stream_begin(scalce_batch *b, int mate, const uint32_t *d_table, const uint8_t *d_symbols,
                                                 uint64_t nsym, void *stream) {
  if (!b || mate < 0 || mate >= b->nm || !d_table || (nsym && !d_symbols) || b->p.no_ac) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  scalce_ctx *c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(b->table[mate].p, d_table, sizeof(u32) * 512000, hipMemcpyDeviceToDevice, s));
  return encode_stream(b, mate, d_symbols, nsym, s);
}
extern "C" int scalce_batch_entropy_stream(scalce_batch *b, int mate, const uint32_t *d_table, const uint8_t *d_symbols,
                                           uint64_t nsym, void *stream) {
  if (!b) return SCALCE_ERR_ARG;
  StageTimer tm(b, ST_ENTROPY, (hipStream_t)stream);
  int rc = scalce_batch_entropy_stream_begin(b, mate, d_table, d_symbols, nsym, stream);
  if (rc) return rc;
  return entropy_collect(b, (hipStream_t)stream);
}

extern "C" int scalce_ac_scale(scalce_ctx *c, const uint64_t *d_counters, uint32_t factor, uint32_t *d_table, void *stream) {
  if (!c || !d_counters || !d_table || !factor) return SCALCE_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  LAUNCH(ac_scale_k, cdiv(512000, 256), 256, 0, (hipStream_t)stream, reinterpret_cast<const u64 *>(d_counters), factor, d_table);
  return SCALCE_OK;
}

// dst[piece_dst[p] + i] = src[piece_src[p] + i] for i < piece_len[p]; pieces sorted by piece_src, contiguous in src
extern "C" int scalce_copy_pieces(scalce_ctx *c, const uint8_t *d_src, uint8_t *d_dst, const uint64_t *d_piece_src,
                                  const uint64_t *d_piece_dst, uint32_t npieces, uint64_t total_bytes, void *stream) {
  if (!c) return SCALCE_ERR_ARG;
  if (!npieces || !total_bytes) return SCALCE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  LAUNCH(copy_pieces_k, cdiv(total_bytes + 15, 256 * 16 * CP_CHUNKS), 256, 0, (hipStream_t)stream, d_src, d_dst,
         reinterpret_cast<const u64 *>(d_piece_src), reinterpret_cast<const u64 *>(d_piece_dst), npieces, (u64)total_bytes);
  return SCALCE_OK;
}

extern "C" int scalce_batch_compress(scalce_batch *b, const uint8_t *t1, uint64_t n1, const uint8_t *t2, uint64_t n2, void *stream) {
  if (!b) return SCALCE_ERR_ARG;
  int rc;
  if ((rc = scalce_batch_ingest(b, 0, t1, n1, stream))) return rc;
  if (b->nm == 2 && (rc = scalce_batch_ingest(b, 1, t2, n2, stream))) return rc;
  if ((rc = scalce_batch_quality(b, stream))) return rc;
  if ((rc = scalce_batch_tokenize(b, nullptr, stream))) return rc;
  if ((rc = scalce_batch_order(b, stream))) return rc;
  if ((rc = scalce_batch_emit(b, stream))) return rc;
  if ((rc = scalce_batch_entropy(b, nullptr, stream))) return rc;
  return SCALCE_OK;
}

// Every stage in front of the entropy coder (ingest .. emit) of a shard that is resident as text, on `stream`.
// (Round 5 ran the quality statistics on a second stream beside the tie-break's windows -- a few hundred launches of ~13 us that
// leave most of the chip idle: 73.97 against 74.04 ms per shard in the bench, as in round 3.  What the windows leave idle the coder
// launches of the other shards in flight already use.  One stream.)
extern "C" int scalce_batch_front(scalce_batch *b, const uint8_t *t1, uint64_t n1, const uint8_t *t2, uint64_t n2, void *stream) {
  if (!b) return SCALCE_ERR_ARG;
  int rc;
  if ((rc = scalce_batch_ingest(b, 0, t1, n1, stream))) return rc;
  if (b->nm == 2 && (rc = scalce_batch_ingest(b, 1, t2, n2, stream))) return rc;
  if ((rc = scalce_batch_quality(b, stream))) return rc;
  if ((rc = scalce_batch_tokenize(b, nullptr, stream))) return rc;
  if ((rc = scalce_batch_order(b, stream))) return rc;
  return scalce_batch_emit(b, stream);
}

extern "C" int scalce_batch_finish(scalce_batch *b, void *stream) {
  if (!b) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(b->ctx, hipSetDevice(b->ctx->device));
  HIP_TRY(b->ctx, hipStreamSynchronize(s));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_err(b->ctx, "kernel launch failed: %s", hipGetErrorString(e)); return SCALCE_ERR_HIP; }
  { int rc = entropy_collect(b, s); if (rc) return rc; }
  return check_device_error(b, s);
}

extern "C" uint64_t scalce_batch_reads(const scalce_batch *b) { return b ? b->N : 0; }
extern "C" int scalce_batch_params(const scalce_batch *b, scalce_params *out) {
  if (!b || !out) return SCALCE_ERR_ARG;
  *out = b->p;
  return SCALCE_OK;
}

// the first two and the last two q' symbols of the rows held (input order), and how many symbols there are: what a rank of a
// sharded run tells its neighbours (the trigrams that straddle a rank boundary) -- without asking for SCALCE_OUT_QINPUT as
// one array, which fused rows would have to be copied together for
extern "C" int scalce_batch_qinput_edges(scalce_batch *b, int mate, uint8_t edge[4], uint64_t *nsym, int32_t *read_len, void *stream) {
  if (!b || mate < 0 || mate >= b->nm || !edge || !nsym) return SCALCE_ERR_ARG;
  scalce_ctx *c = b->ctx;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(c, hipSetDevice(c->device));
  const u64 L = (u64)b->L[mate], n = b->N * L, QS = b->qstride[mate];
  *nsym = n;
  if (read_len) *read_len = b->L[mate];
  edge[0] = edge[1] = edge[2] = edge[3] = 0;
  const u8 *q = b->q[mate].as<u8>();
  auto at = [&](u64 t) { return q + (t / L) * QS + (t % L); };
  // (on the caller's stream, behind the ingest: a device-wide wait would sit behind every coder that is running)
  if (n >= 2) {
    HIP_TRY(c, hipMemcpyAsync(&edge[0], at(0), 1, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&edge[1], at(1), 1, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&edge[2], at(n - 2), 1, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&edge[3], at(n - 1), 1, hipMemcpyDeviceToHost, s));
  } else if (n == 1) {
    HIP_TRY(c, hipMemcpyAsync(&edge[0], at(0), 1, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(c, hipStreamSynchronize(s));
  if (n == 1) edge[3] = edge[0];
  return SCALCE_OK;
}

extern "C" int scalce_batch_set_frame_on_demand(scalce_batch *b, int on) {
  if (!b) return SCALCE_ERR_ARG;
  b->frame_on_demand = on != 0;
  return SCALCE_OK;
}

extern "C" int scalce_batch_qual_bytes(const scalce_batch *b, int mate, uint64_t *nbytes) {
  if (!b || !nbytes || mate < 0 || mate >= b->nm) return SCALCE_ERR_ARG;
  *nbytes = b->out_qual_bytes[mate];
  return SCALCE_OK;
}

// bytes [offset, offset + nbytes) of mate's framed quality stream -> dst (device memory, or pinned host memory: the
// kernel's stores go over the link), straight from the coder's block buffers when the frames are only laid out
extern "C" int scalce_batch_qual_window(scalce_batch *b, int mate, uint64_t offset, uint64_t nbytes, void *dst, void *stream) {
  if (!b || mate < 0 || mate >= b->nm || (nbytes && !dst)) return SCALCE_ERR_ARG;
  scalce_ctx *c = b->ctx;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(c, hipSetDevice(c->device));
  if (b->ent_pending[mate] || b->frame_deferred[mate]) { set_err(c, "collect the entropy stage first (scalce_batch_finish)"); return SCALCE_ERR_ARG; }
  if (offset > b->out_qual_bytes[mate] || nbytes > b->out_qual_bytes[mate] - offset) { set_err(c, "window beyond the stream"); return SCALCE_ERR_ARG; }
  if (!nbytes) return SCALCE_OK;
  if (b->p.no_ac || !b->frame_virtual[mate]) {  // the stream exists as such
    const u8 *src = b->p.no_ac ? b->qs(mate).as<u8>() : b->out_qual[mate].as<u8>();
    HIP_TRY(c, hipMemcpyAsync(dst, src + offset, nbytes, hipMemcpyDefault, s));
    return SCALCE_OK;
  }
  if ((uintptr_t)dst & 3) { set_err(c, "window destination must be 4-byte aligned"); return SCALCE_ERR_ARG; }
  // the blocks whose frames meet the window
  const std::vector<u64> &off = b->frame_off_host[mate];
  const u32 nblk = b->frame_virtual[mate];
  if (off.size() != (size_t)nblk) { set_err(c, "internal: frame layout not collected"); return SCALCE_ERR_ARG; }
  const u32 b0 = (u32)(std::upper_bound(off.begin(), off.end(), (u64)offset) - off.begin()) - 1u;  // off[0] = 0 <= offset
  const u32 b1 = (u32)(std::lower_bound(off.begin(), off.end(), (u64)(offset + nbytes)) - off.begin());
  LAUNCH(ac_frame_window_k, dim3(cdiv(b->ac_stride[mate], 16 * 256), b1 - b0), 256, 0, s, b->ac_base[mate], b->ac_stride[mate],
         b->ac_sizes[mate].as<u32>(), b->ac_off[mate].as<u64>(), (u64)offset, (u64)(offset + nbytes), static_cast<u8 *>(dst), b0);
  return launch_failed(c);
}

// SCALCE_OUT_QUAL as one device buffer for callers that want that: the frames laid out by entropy_collect are copied now
static int materialize_frames(scalce_batch *b, int m) {
  if (!b->frame_virtual[m]) return SCALCE_OK;
  scalce_ctx *c = b->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipDeviceSynchronize());
  ENSURE(b, b->out_qual[m], (size_t)b->out_qual_bytes[m] + 64);
  LAUNCH(ac_frame_k, dim3(cdiv(b->ac_stride[m], 16 * 256), b->frame_virtual[m]), 256, 0, (hipStream_t) nullptr, b->ac_base[m], b->ac_stride[m],
         b->ac_sizes[m].as<u32>(), b->ac_off[m].as<u64>(), b->out_qual[m].as<u8>());
  HIP_TRY(c, hipDeviceSynchronize());
  b->frame_virtual[m] = 0;
  return launch_failed(c);
}

extern "C" int scalce_batch_output(const scalce_batch *b, int which, int mate, const void **d_ptr, uint64_t *nbytes) {
  if (!b || !d_ptr || !nbytes || mate < 0 || mate >= b->nm) return SCALCE_ERR_ARG;
  const u32 nb1 = (u32)b->ctx->A.n_buckets + 1;
  if (which == SCALCE_OUT_QUAL && !b->p.no_ac && b->frame_virtual[mate]) {
    int rc = materialize_frames(const_cast<scalce_batch *>(b), mate);
    if (rc) return rc;
  }
  switch (which) {
    case SCALCE_OUT_READS: *d_ptr = b->out_reads[mate].p; *nbytes = b->out_reads_bytes[mate]; break;
    case SCALCE_OUT_NAMES: *d_ptr = b->out_names.p; *nbytes = b->out_names_bytes; break;
    case SCALCE_OUT_QUAL:
      *d_ptr = b->p.no_ac ? b->qs(mate).p : b->out_qual[mate].p;
      *nbytes = b->out_qual_bytes[mate];
      break;
    case SCALCE_OUT_TABLE: *d_ptr = b->table[mate].p; *nbytes = sizeof(u32) * 512000; break;
    case SCALCE_OUT_FREQ4: *d_ptr = b->freq4[mate].p; *nbytes = sizeof(u64) * 512000; break;
    case SCALCE_OUT_TOKENS: *d_ptr = b->tokens.p; *nbytes = sizeof(int32_t) * 2 * b->N; break;
    case SCALCE_OUT_PERM: *d_ptr = b->perm; *nbytes = sizeof(u32) * b->N; break;
    case SCALCE_OUT_QSTREAM: *d_ptr = b->qs(mate).p; *nbytes = b->N * (u64)b->L[mate]; break;
    case SCALCE_OUT_BUCKET_COUNTS: *d_ptr = b->tok_open ? b->counts.p : b->counts_total.p; *nbytes = sizeof(u64) * nb1; break;
    case SCALCE_OUT_QINPUT:
      *nbytes = b->N * (u64)b->L[mate];
      if (b->qstride[mate] == (u32)b->L[mate]) { *d_ptr = b->q[mate].p; break; }
      {  // fused rows: the q' of every row as one array, put together on request
        scalce_batch *mb = const_cast<scalce_batch *>(b);
        if (hipSetDevice(b->ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return SCALCE_ERR_HIP;
        int rc = ensure(mb, mb->q_compact, (size_t)*nbytes + 64);
        if (rc) return rc;
        if (b->N) LAUNCH(compact_q_k, 4096, 256, 0, (hipStream_t) nullptr, b->N, b->q[mate].as<u8>(), b->qstride[mate], (u32)b->L[mate], mb->q_compact.as<u8>());
        if (hipDeviceSynchronize() != hipSuccess) return SCALCE_ERR_HIP;
        *d_ptr = mb->q_compact.p;
      }
      break;
    case SCALCE_OUT_NAMELEN: *d_ptr = b->namelen.p; *nbytes = b->N; break;
    case SCALCE_OUT_BUCKET_NAME_BYTES: *d_ptr = b->bucket_name_bytes.p; *nbytes = b->bucket_name_bytes.p ? sizeof(u64) * nb1 : 0; break;
    default: return SCALCE_ERR_ARG;
  }
  return SCALCE_OK;
}

extern "C" int scalce_batch_stage_ms(scalce_batch *b, int which, float *ms, int *launches) {
  if (!b || which < 0 || which >= ST_COUNT) return SCALCE_ERR_ARG;
  if (ms) *ms = b->stage_ms[which];
  if (launches) *launches = b->stage_launches[which];
  return SCALCE_OK;
}
extern "C" void scalce_batch_stage_reset(scalce_batch *b, int enable) {
  if (!b) return;
  b->timing = enable != 0;
  for (int i = 0; i < ST_COUNT; i++) { b->stage_ms[i] = 0; b->stage_launches[i] = 0; }
}

extern "C" void scalce_batch_kernel_timing(scalce_batch *b, int enable) {
  if (!b) return;
  b->ktiming = enable != 0;
  b->kev_used = 0;
  b->k_in_bytes = b->k_out_bytes = 0;
}
extern "C" int scalce_batch_kernel_ms(scalce_batch *b, double *total_ms, int *launches, uint64_t *bytes_in,
                                      uint64_t *bytes_out) {
  if (!b) return SCALCE_ERR_ARG;
  double tot = 0;
  for (size_t i = 0; i < b->kev_used; i++) {
    float ms = 0;
    HIP_TRY(b->ctx, hipEventSynchronize(b->kev[i].second));
    HIP_TRY(b->ctx, hipEventElapsedTime(&ms, b->kev[i].first, b->kev[i].second));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int)b->kev_used;
  if (bytes_in) *bytes_in = b->k_in_bytes;
  if (bytes_out) *bytes_out = b->k_out_bytes;
  return SCALCE_OK;
}

extern "C" int scalce_memcpy_d2h(scalce_ctx *c, void *dst, const void *src, uint64_t n) {
  if (!c) return SCALCE_ERR_ARG;
  if (!n) return SCALCE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
  return SCALCE_OK;
}
extern "C" int scalce_memcpy_h2d(scalce_ctx *c, void *dst, const void *src, uint64_t n) {
  if (!c) return SCALCE_ERR_ARG;
  if (!n) return SCALCE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpy(dst, src, n, hipMemcpyHostToDevice));
  return SCALCE_OK;
}
extern "C" int scalce_memcpy_d2d(scalce_ctx *c, void *dst, const void *src, uint64_t n, void *stream) {
  if (!c) return SCALCE_ERR_ARG;
  if (!n) return SCALCE_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SCALCE_OK;
}
extern "C" int scalce_batch_stats(const scalce_batch *b, uint32_t out[6]) {
  if (!b || !out) return SCALCE_ERR_ARG;
  out[0] = b->ntie; out[1] = b->nev; out[2] = b->jacobi_iters; out[3] = b->nchunks;
  out[4] = b->order_run_members;
  out[5] = b->tie_fallback ? 1u : 0u;
  return SCALCE_OK;
}

extern "C" int scalce_selftest_ac(scalce_ctx *c, uint64_t ncases, uint32_t seed, int general, uint32_t out[6]) {
  if (!c || !out) return SCALCE_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  u32 *d = nullptr;
  HIP_TRY(c, hipMalloc(&d, 8 * sizeof(u32)));
  HIP_TRY(c, hipMemset(d, 0, 8 * sizeof(u32)));
  LAUNCH(ac_selftest_k, cdiv(ncases, 256), 256, 0, 0, (u64)ncases, seed, general, d);
  HIP_TRY(c, hipDeviceSynchronize());
  HIP_TRY(c, hipMemcpy(out, d, 6 * sizeof(u32), hipMemcpyDeviceToHost));
  hipFree(d);
  return SCALCE_OK;
}

// ---- decode ---------------------------------------------------------------------------------------------
extern "C" int scalce_ac_decode(scalce_ctx *c, const uint32_t *table_host, const uint8_t *d_blocks, uint64_t nbytes,
                                uint64_t nsym, uint8_t *d_out, void *stream) {
  if (!c || !table_host || !d_blocks || !d_out) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(c, hipSetDevice(c->device));
  const u32 nblk = cdiv(nsym, AC_BLOCK_SYMS);
  if (!nblk) return SCALCE_OK;
  // walk the [u32 size][bytes] frames: the walk is serial by nature (each size says where the next one is), so a small
  // device kernel follows the chain once and the host takes all offsets with one copy (a blocking 4-byte copy per block was
  // 477 round trips for a 50 M-read shard, 5 724 for 200 M pairs)
  std::vector<u64> off(nblk);
  std::vector<u32> sz(nblk);
  {
    u64 *d_walk = nullptr;
    HIP_TRY(c, hipMalloc(&d_walk, sizeof(u64) * ((size_t)nblk * 2 + 2)));
    LAUNCH(ac_frame_walk_k, 1, 1, 0, s, d_blocks, (u64)nbytes, nblk, d_walk);
    std::vector<u64> w((size_t)nblk * 2 + 2);
    hipError_t e = hipMemcpyAsync(w.data(), d_walk, sizeof(u64) * w.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(d_walk);
    if (e != hipSuccess) { set_err(c, "reading the block frames: %s", hipGetErrorString(e)); return SCALCE_ERR_HIP; }
    if (w[(size_t)nblk * 2] != 0) { set_err(c, "(ERROR) truncated quality stream"); return SCALCE_ERR_FORMAT; }
    for (u32 i = 0; i < nblk; i++) { off[i] = w[2 * (size_t)i]; sz[i] = (u32)w[2 * (size_t)i + 1]; }
  }
  u32 *d_table = nullptr, *d_cum = nullptr, *d_sz = nullptr;
  uint4 *d_tab = nullptr;
  u64 *d_off = nullptr;
  HIP_TRY(c, hipMalloc(&d_table, sizeof(u32) * 512000));
  HIP_TRY(c, hipMalloc(&d_cum, sizeof(u32) * 6400 * 81));
  HIP_TRY(c, hipMalloc(&d_tab, sizeof(uint4) * 512000));
  HIP_TRY(c, hipMalloc(&d_off, sizeof(u64) * nblk));
  HIP_TRY(c, hipMalloc(&d_sz, sizeof(u32) * nblk));
  HIP_TRY(c, hipMemcpyAsync(d_table, table_host, sizeof(u32) * 512000, hipMemcpyHostToDevice, s));
  HIP_TRY(c, hipMemcpyAsync(d_off, off.data(), sizeof(u64) * nblk, hipMemcpyHostToDevice, s));
  HIP_TRY(c, hipMemcpyAsync(d_sz, sz.data(), sizeof(u32) * nblk, hipMemcpyHostToDevice, s));
  LAUNCH(ac_table_k, cdiv(6400, 64), 64, 0, s, d_table, d_tab, d_cum, (u32 *)nullptr);
  AcDecArgs a;
  a.in = d_blocks; a.blk_off = d_off; a.blk_size = d_sz; a.nsym = nsym; a.tab = d_tab; a.out = d_out;
  // span of the symbols that occur (scaled count > 1 in some context) and their totals: what the compact rows hold and
  // which contexts go to LDS.  (A symbol outside the span can still be coded -- one occurrence scales down to the
  // floor count 1 -- and takes the full-row path in the kernel.)
  u32 smin = AC_D, smax = 0;
  std::vector<u64> tot(AC_D, 0);
  for (u32 ctx = 0; ctx < 6400; ctx++)
    for (u32 sy = 0; sy < AC_D; sy++) {
      const u32 v = table_host[(size_t)ctx * AC_D + sy];
      if (v > 1) { smin = std::min(smin, sy); smax = std::max(smax, sy); tot[sy] += v; }
    }
  uint2 *d_rows = nullptr;
  const char *wpb_env = getenv("SCALCE_AC_DECODE_WPB");  // test hook: chains per workgroup (2, 4, 8, 16); 0 = the plain decoder
  const bool cached = smin <= smax && smax - smin + 2 <= 64 && !(wpb_env && atoi(wpb_env) == 0);
  if (cached) {
    AcDecCachedArgs ca;
    memset(&ca, 0, sizeof ca);
    ca.d = a;
    ca.smin = smin;
    ca.S1 = smax - smin + 2;
    ca.nblk = nblk;
    std::vector<u32> order;
    for (u32 sy = smin; sy <= smax; sy++) if (tot[sy]) order.push_back(sy);
    std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) { return tot[x] > tot[y]; });
    u32 W = 1;
    while (W < 32 && W < order.size() && (u64)(W + 1) * (W + 1) * ca.S1 <= AC_DEC_CACHE_ENTRIES) W++;
    ca.W = W;
    memset(ca.rank, 0xFF, sizeof ca.rank);
    for (u32 r = 0; r < W; r++) { ca.hot[r] = (u8)order[r]; ca.rank[order[r]] = (u8)r; }
    HIP_TRY(c, hipMalloc(&d_rows, sizeof(uint2) * (6400 * ca.S1 + 64)));  // (+ 64: ac_decode_fast_k reads a row with all lanes)
    HIP_TRY(c, hipMemsetAsync(d_rows + 6400 * (size_t)ca.S1, 0, sizeof(uint2) * 64, s));
    LAUNCH(ac_dec_rows_k, cdiv(6400u * ca.S1, 256), 256, 0, s, d_tab, smin, ca.S1, d_rows);
    ca.rows = d_rows;
    // Waves of a workgroup share the LDS cache of hot rows (one workgroup per CU): two chains per workgroup keep the
    // latency of a block lowest; from 512 blocks on, eight per workgroup -- two chains per SIMD interleave their issue
    // slots -- put four times as many blocks in flight.
    // ONE cached decoder (round 5; rounds 2-4 kept four): ac_decode_tight_k, the loop written by hand for the scalar unit.  It
    // needs what every table of quality strings gives -- no symbol 79 among those that occur (that symbol marks "last of its
    // context" in the rows) and no context total above 2^29 (as for the encoder's plain step); any other table takes the plain
    // decoder below, the reference's own loop (arithmetic.cpp:196-268) a wavefront per block.
    u64 max_total = 0;
    for (u32 ctx = 0; ctx < 6400; ctx++) {
      u64 t = 0;
      for (u32 sy = 0; sy < AC_D; sy++) t += table_host[(size_t)ctx * AC_D + sy];
      max_total = std::max(max_total, t);
    }
    if (smax < AC_D - 1 && max_total <= (1ull << 29)) {
      int wpb = nblk <= 512 ? 2 : nblk <= 1024 ? 4 : nblk <= 2048 ? 8 : 16;  // 16 = four chains per SIMD: a chain issues one instruction in five cycles
      if (wpb_env) wpb = atoi(wpb_env);
      if (wpb == 2) LAUNCH(ac_decode_tight_k<2>, cdiv(nblk, 2), 128, 0, s, ca);
      else if (wpb == 4) LAUNCH(ac_decode_tight_k<4>, cdiv(nblk, 4), 256, 0, s, ca);
      else if (wpb == 16) LAUNCH(ac_decode_tight_k<16>, cdiv(nblk, 16), 1024, 0, s, ca);
      else LAUNCH(ac_decode_tight_k<8>, cdiv(nblk, 8), 512, 0, s, ca);
    } else {
      LAUNCH(ac_decode_k, nblk, 64, 0, s, a);
    }
  } else {
    LAUNCH(ac_decode_k, nblk, 64, 0, s, a);
  }
  HIP_TRY(c, hipStreamSynchronize(s));
  hipFree(d_table); hipFree(d_cum); hipFree(d_tab); hipFree(d_off); hipFree(d_sz);
  if (d_rows) hipFree(d_rows);
  return SCALCE_OK;
}

// ---- decode side, records -> FASTQ text (decompress.cpp:240-366) -----------------------------------
extern "C" uint64_t scalce_fastq_text_bytes(int read_len, uint64_t nrecords, uint64_t names_bytes, const char *library) {
  const u64 L = (u64)read_len, N = nrecords;
  if (!library) return names_bytes - N + N * (2 * L + 6);  // names_bytes = sum of (1 + n)
  u64 digits = N, p = 10;  // digits of 0 .. N-1
  for (int t = 2; t <= 20 && N > p; t++, p *= 10) digits += N - p;
  return N * (strlen(library) + 2 * L + 7) + digits;
}

extern "C" int scalce_fastq_records(scalce_ctx *c, int read_len, int has_buckets, const uint8_t *reads_host, uint64_t reads_bytes,
                                    uint64_t nrecords, const uint8_t *d_qual, int64_t phred_offset, const uint8_t *names_host,
                                    uint64_t names_bytes, const char *library, int mate_digit, uint8_t *d_out, uint64_t out_cap,
                                    uint64_t *out_bytes, uint64_t *record_offsets_host, void *stream) {
  if (!c || read_len <= 0 || !reads_host || (!names_host && !library) || !d_out || (nrecords && !d_qual)) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(c, hipSetDevice(c->device));
  const u32 L = (u32)read_len;
  const u32 sz_meta = has_buckets ? (L > 255 ? 2u : 1u) : 0u;
  // 1. the bucket directory (decompress.cpp:262-270): headers sit between the buckets' records, so the walk is serial
  std::vector<FqBucket> dir;
  if (has_buckets) {
    u64 pos = 0, k = 0;
    while (pos + 12 <= reads_bytes) {
      int32_t core;
      u64 cnt;
      memcpy(&core, reads_host + pos, 4);
      memcpy(&cnt, reads_host + pos + 4, 8);
      pos += 12;
      FqBucket b;
      memset(&b, 0, sizeof b);
      if (core != SCALCE_ROOT_CORE) {
        if (core < 0 || core >= (int)c->A.patterns.size()) {
          set_err(c, "(ERROR) archive refers to core %d which the core table does not have", core);
          return SCALCE_ERR_FORMAT;
        }
        const std::string &cs = c->A.patterns[core];
        if (cs.size() > sizeof b.core || cs.size() > L) { set_err(c, "(ERROR) core %d does not fit the reads", core); return SCALCE_ERR_FORMAT; }
        b.core_len = (u32)cs.size();
        memcpy(b.core, cs.data(), cs.size());
      }
      b.first = k;
      b.off = pos;
      b.rec_bytes = (L - b.core_len + 3) / 4 + sz_meta;
      if (cnt > (reads_bytes - pos) / b.rec_bytes) { set_err(c, "(ERROR) truncated read stream"); return SCALCE_ERR_FORMAT; }
      pos += cnt * b.rec_bytes;
      k += cnt;
      if (cnt) dir.push_back(b);
    }
    if (k != nrecords) {
      set_err(c, "(ERROR) the read stream holds %llu records, the quality stream %llu", (unsigned long long)k, (unsigned long long)nrecords);
      return SCALCE_ERR_FORMAT;
    }
  } else {
    FqBucket b;
    memset(&b, 0, sizeof b);
    b.rec_bytes = (L + 3) / 4;
    if (nrecords > reads_bytes / b.rec_bytes) { set_err(c, "(ERROR) truncated read stream"); return SCALCE_ERR_FORMAT; }
    dir.push_back(b);
  }
  // 2. where every name starts (each length byte says where the next one is: serial as well)
  std::vector<u64> name_off;
  if (names_host) {
    name_off.resize(nrecords + 1);
    u64 pos = 0;
    for (u64 k = 0; k < nrecords; k++) {
      if (pos >= names_bytes) { set_err(c, "(ERROR) truncated name stream"); return SCALCE_ERR_FORMAT; }
      name_off[k] = pos;
      pos += 1 + (u64)names_host[pos];
    }
    if (pos > names_bytes) { set_err(c, "(ERROR) truncated name stream"); return SCALCE_ERR_FORMAT; }
    name_off[nrecords] = pos;
    names_bytes = pos;
  }
  const u64 total = scalce_fastq_text_bytes(read_len, nrecords, names_bytes, names_host ? nullptr : library);
  if (out_bytes) *out_bytes = total;
  if (total > out_cap) { set_err(c, "output buffer of %llu bytes, the text needs %llu", (unsigned long long)out_cap, (unsigned long long)total); return SCALCE_ERR_CAPACITY; }
  if (!nrecords) return SCALCE_OK;
  FqArgs a;
  memset(&a, 0, sizeof a);
  u8 *d_reads = nullptr, *d_names = nullptr;
  FqBucket *d_dir = nullptr;
  u64 *d_noff = nullptr, *d_roff = nullptr;
  auto release = [&]() { hipFree(d_reads); hipFree(d_names); hipFree(d_dir); hipFree(d_noff); hipFree(d_roff); };
#define FQ_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { release(); set_err(c, "%s failed: %s", #expr, hipGetErrorString(e_)); return SCALCE_ERR_HIP; } } while (0)
  FQ_TRY(hipMalloc(&d_reads, reads_bytes + 64));
  FQ_TRY(hipMalloc(&d_dir, sizeof(FqBucket) * dir.size()));
  FQ_TRY(hipMemcpyAsync(d_reads, reads_host, reads_bytes, hipMemcpyHostToDevice, s));
  FQ_TRY(hipMemcpyAsync(d_dir, dir.data(), sizeof(FqBucket) * dir.size(), hipMemcpyHostToDevice, s));
  if (names_host) {
    FQ_TRY(hipMalloc(&d_names, names_bytes + 64));
    FQ_TRY(hipMalloc(&d_noff, sizeof(u64) * (nrecords + 1)));
    FQ_TRY(hipMemcpyAsync(d_names, names_host, names_bytes, hipMemcpyHostToDevice, s));
    FQ_TRY(hipMemcpyAsync(d_noff, name_off.data(), sizeof(u64) * (nrecords + 1), hipMemcpyHostToDevice, s));
  } else {
    a.lib_len = (u32)std::min<size_t>(strlen(library), sizeof a.lib - 1);
    if (strlen(library) >= sizeof a.lib) { release(); set_err(c, "library name longer than %zu characters", sizeof a.lib - 1); return SCALCE_ERR_ARG; }
    memcpy(a.lib, library, a.lib_len);
  }
  if (record_offsets_host) FQ_TRY(hipMalloc(&d_roff, sizeof(u64) * (nrecords + 1)));
  a.reads = d_reads; a.dir = d_dir; a.nbuckets = (u32)dir.size(); a.nrecords = nrecords; a.L = L; a.sz_meta = sz_meta;
  a.qual = d_qual; a.phred = (u32)phred_offset; a.names = d_names; a.name_off = d_noff;
  a.mate_digit = (u32)mate_digit; a.out = d_out; a.rec_off = d_roff;
  const u64 waves = (nrecords + FQ_RECORDS_PER_WAVE - 1) / FQ_RECORDS_PER_WAVE;
  LAUNCH(fastq_records_k, cdiv(waves, 4), 256, 0, s, a);
  if (record_offsets_host)
    FQ_TRY(hipMemcpyAsync(record_offsets_host, d_roff, sizeof(u64) * (nrecords + 1), hipMemcpyDeviceToHost, s));
  FQ_TRY(hipStreamSynchronize(s));
  FQ_TRY(hipGetLastError());
#undef FQ_TRY
  release();
  return SCALCE_OK;
}
