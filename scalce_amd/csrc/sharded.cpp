// sharded.cpp -- one archive from a read stream split over several GPUs (one process per GPU, RCCL over xGMI).
//
// Every rank holds a contiguous piece of the input (rank order = input order).  What the reference carries across reads
// is exchanged, nothing else (sizes for 8 ranks x 50 M reads of 100 bp):
//   spill rule      compress.cpp:702-715  running record bytes of the open chunk: a chain of `world` 8-byte messages, then
//                                         an all-gather of the cuts; rank boundaries move to the nearest cut and the records
//                                         in between change owner AS TEXT (at most half a chunk per boundary, ~2.6 GB); the
//                                         rows that stay are kept as they are (scalce_batch_rewindow)
//   quality model   qualities.cpp:179-198 all-reduce of the 80^3 counters (4 MB per mate) + the trigrams that straddle a
//                                         rank boundary (all-gather of 4 edge symbols)
//   tie-break       reads.cpp:246,420     per round an all-gather of the per-bucket counts (125 KB); a rank's prior is the
//                                         sum over the ranks before it; several local sweeps per round
//   bucket layout   compress.cpp:364-379  all-gather of final counts and name bytes per bucket (2 x 125 KB)
//   coder blocks    arithmetic.cpp:318-363 cut every 10 MiB of the RUN-WIDE reordered stream: one all-to-all of q' bytes
//                                         into contiguous block ranges (~ L bytes per read, 7/8 of it leaves the rank)
// With rank boundaries ON chunk boundaries, "inside a bucket: rank 0's records, rank 1's, ..." is exactly the merge order
// of compress.cpp:104-159, so the archive equals the one-GPU archive and the reference's at -T 1 with the same -B.  A run
// that -B does not cut at all is one chunk: its rows all go to rank 0 (scalce_shard_plan_boundaries).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/scalce_hip.h"

namespace {

typedef uint64_t u64;
constexpr u64 AC_BLOCK = 10ull * 1024 * 1024;

__global__ void add_counts_into_k(uint32_t nb1, const u64 *mine, u64 *acc) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb1) acc[b] += mine[b];
}
__global__ void add_ones_k(u64 *table, const uint32_t *keys, uint32_t n) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    for (uint32_t i = 0; i < n; i++) table[keys[i]] += 1;
}

struct Fail {
  std::string msg;
  int rc;
};

}  // namespace

#define SH_RC(ctx, expr)                                                        \
  do {                                                                          \
    int rc_ = (expr);                                                           \
    if (rc_) throw Fail{std::string(#expr) + ": " + scalce_last_error(ctx), rc_}; \
  } while (0)
#define SH_CM(comm, expr)                                                        \
  do {                                                                           \
    int rc_ = (expr);                                                            \
    if (rc_) throw Fail{std::string(#expr) + ": " + scalce_comm_error(comm), rc_}; \
  } while (0)
#define SH_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) throw Fail{std::string(#expr) + ": " + hipGetErrorString(e_), SCALCE_ERR_HIP}; \
  } while (0)

namespace {

// Scratch of a call: grow-only device buffers that persist in the process (one rank = one process), handed out in call
// order -- hipMalloc / hipFree synchronise the whole device, and a sharded call usually runs beside the coder of the
// previous shards on another stream: an allocation per call made every call wait for that coder.
struct Arena {
  void *p = nullptr;
  size_t cap = 0;
};
// (per host thread: a caller that drives two pipelines from two threads -- each with a communicator of its own, so that one's
//  waits overlap the other's work -- gives each its own scratch)
static thread_local std::vector<Arena> g_scratch;
struct DevMem {
  size_t next = 0;
  template <typename T> T *alloc(size_t n) {
    const size_t bytes = (n ? n : 1) * sizeof(T) + 64;
    if (next == g_scratch.size()) g_scratch.emplace_back();
    Arena &a = g_scratch[next++];
    if (a.cap < bytes) {
      if (a.p) hipFree(a.p);
      a.p = nullptr;
      a.cap = 0;
      const size_t want = bytes + bytes / 8;
      SH_HIP(hipMalloc(&a.p, want));
      a.cap = want;
    }
    return static_cast<T *>(a.p);
  }
};
// what the coder still reads when the call has returned lives in the RESULT (reused when the caller passes the result of
// an earlier call on the same batch slot back in)
template <typename T> T *keep_alloc(scalce_shard_result *res, int slot, size_t n) {
  const size_t bytes = (n ? n : 1) * sizeof(T) + 64;
  if (res->keep[slot] && res->keep_bytes[slot] >= bytes) return static_cast<T *>(res->keep[slot]);
  if (res->keep[slot]) hipFree(res->keep[slot]);
  res->keep[slot] = nullptr;
  res->keep_bytes[slot] = 0;
  const size_t want = bytes + bytes / 8;
  SH_HIP(hipMalloc(&res->keep[slot], want));
  res->keep_bytes[slot] = want;
  return static_cast<T *>(res->keep[slot]);
}

template <typename T> std::vector<T> gather_host(scalce_comm *comm, const T *mine, size_t count, T *d_send, T *d_recv, hipStream_t s) {
  const int W = scalce_comm_world(comm);
  SH_HIP(hipMemcpyAsync(d_send, mine, count * sizeof(T), hipMemcpyHostToDevice, s));
  SH_CM(comm, scalce_comm_all_gather(comm, d_send, d_recv, count * sizeof(T), s));
  std::vector<T> all((size_t)W * count);
  SH_HIP(hipMemcpyAsync(all.data(), d_recv, all.size() * sizeof(T), hipMemcpyDeviceToHost, s));
  SH_HIP(hipStreamSynchronize(s));
  return all;
}

}  // namespace

// Host-only plan math, exported so that it can be checked without a GPU -------------------------------------------------
// Rank boundaries g[0..world] (run-wide row of every rank's first record, g[world] = rows of the run) move to the nearest
// cut of the run-wide -B rule; gn[] = the moved boundaries (non-decreasing; ranks may end up without records when chunks are
// larger than a rank's share).  A run without any cut is ONE chunk -- less than -B bytes of records, what the reference keeps
// in memory as one bucket set (compress.cpp:708-715) and sorts bucket by bucket as a whole (reads.cpp:547-634): all its rows
// go to rank 0, whose order and emit stages then ARE the merge of every bucket across the ranks (round 4; rounds 1-3 refused
// such a run with SCALCE_ERR_UNCUT).  The other ranks keep their share of the quality statistics and of the coder's blocks.
extern "C" int scalce_shard_plan_boundaries(int world, const uint64_t *g, const uint64_t *cuts_sorted, uint64_t ncuts, uint64_t *gn) {
  if (world < 1 || !g || !gn || (ncuts && !cuts_sorted)) return SCALCE_ERR_ARG;
  for (int r = 0; r <= world; r++) gn[r] = g[r];
  if (!ncuts) {
    for (int r = 1; r < world; r++) gn[r] = g[world];
    return SCALCE_OK;
  }
  for (int r = 1; r < world; r++) {
    const uint64_t *end = cuts_sorted + ncuts, *it = std::lower_bound(cuts_sorted, end, g[r]);
    uint64_t best = it == end ? cuts_sorted[ncuts - 1] : *it;
    if (it != cuts_sorted) { const uint64_t lo = *(it - 1); if (best < g[r] || g[r] - lo <= best - g[r]) best = lo; }
    gn[r] = best;
  }
  for (int r = 1; r < world; r++)
    if (gn[r] < gn[r - 1]) gn[r] = gn[r - 1];
  return SCALCE_OK;
}
// The run-wide reordered quality stream of one mate (buckets in emission order, ranks in order inside a bucket; C[r][k] reads
// of L symbols) dealt out in contiguous ranges of whole 10 MiB blocks, range d to rank d: what `rank` sends to / receives from
// everyone (bytes), its own range [lo, hi), and the pieces of what it receives (source-major, bucket order: contiguous in
// the receive buffer; piece_dst relative to lo).  Piece arrays hold up to world * nb1 entries.
extern "C" int scalce_shard_plan_blocks(int world, int rank, uint32_t nb1, const uint64_t *C, uint64_t L, uint64_t *send_bytes,
                                        uint64_t *recv_bytes, uint64_t *lo_out, uint64_t *hi_out, uint64_t *piece_src, uint64_t *piece_dst,
                                        uint64_t *npieces) {
  if (world < 1 || rank < 0 || rank >= world || !C || !send_bytes || !recv_bytes || !lo_out || !hi_out) return SCALCE_ERR_ARG;
  std::vector<u64> Cg(nb1, 0);
  u64 total_reads = 0;
  for (int r = 0; r < world; r++)
    for (uint32_t k = 0; k < nb1; k++) { Cg[k] += C[(size_t)r * nb1 + k]; total_reads += C[(size_t)r * nb1 + k]; }
  const u64 total = total_reads * L, nblk = (total + AC_BLOCK - 1) / AC_BLOCK;
  auto lo_of = [&](int d) { return std::min<u64>(total, ((u64)d * nblk / world) * AC_BLOCK); };
  std::vector<u64> g0((size_t)world * nb1);  // run-wide start of every (rank, bucket) piece
  {
    u64 base = 0;
    for (uint32_t k = 0; k < nb1; k++) {
      u64 at = base;
      for (int r = 0; r < world; r++) { g0[(size_t)r * nb1 + k] = at; at += C[(size_t)r * nb1 + k] * L; }
      base += Cg[k] * L;
    }
  }
  auto below = [&](int r, u64 X) {  // bytes of rank r's local stream that lie in front of run-wide offset X
    u64 sum = 0;
    for (uint32_t k = 0; k < nb1; k++) {
      const u64 a = g0[(size_t)r * nb1 + k], len = C[(size_t)r * nb1 + k] * L;
      sum += X <= a ? 0 : (X - a < len ? X - a : len);
    }
    return sum;
  };
  const u64 lo = lo_of(rank), hi = lo_of(rank + 1);
  for (int d = 0; d < world; d++) send_bytes[d] = below(rank, lo_of(d + 1)) - below(rank, lo_of(d));
  for (int src = 0; src < world; src++) recv_bytes[src] = below(src, hi) - below(src, lo);
  u64 at = 0, np = 0;
  for (int src = 0; src < world; src++)
    for (uint32_t k = 0; k < nb1; k++) {
      const u64 a0 = g0[(size_t)src * nb1 + k], a1 = a0 + C[(size_t)src * nb1 + k] * L;
      const u64 x = std::max(a0, lo), y = std::min(a1, hi);
      if (y > x) {
        if (piece_src) { piece_src[np] = at; piece_dst[np] = x - lo; }
        np++;
        at += y - x;
      }
    }
  u64 recv_total = 0;
  for (int r = 0; r < world; r++) recv_total += recv_bytes[r];
  if (at != recv_total) return SCALCE_ERR_ARG;
  *lo_out = lo;
  *hi_out = hi;
  if (npieces) *npieces = np;
  return SCALCE_OK;
}

extern "C" void scalce_shard_result_free(scalce_shard_result *r) {
  if (!r || r->magic != 0x5CA1CE5Du) return;
  free(r->counts);
  free(r->name_bytes);
  r->counts = r->name_bytes = nullptr;
  for (int i = 0; i < 8; i++) { if (r->keep[i]) hipFree(r->keep[i]); r->keep[i] = nullptr; r->keep_bytes[i] = 0; }
  r->magic = 0;
}

extern "C" int scalce_sharded_compress(scalce_comm *comm, scalce_ctx *ctx, scalce_batch *b, const uint8_t *d_text1, uint64_t n1,
                                       const uint8_t *d_text2, uint64_t n2, int flags, void *stream, void *coder_stream,
                                       scalce_shard_result *res) {
  if (!comm || !ctx || !b || !res) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int W = scalce_comm_world(comm), rank = scalce_comm_rank(comm);
  if (W > 64) return SCALCE_ERR_ARG;
  {  // a result of an earlier call on this slot keeps its buffers; everything else starts from zero
    scalce_shard_result keep = *res;
    const bool reuse = res->magic == 0x5CA1CE5Du && res->world == W && res->nb1 == (uint32_t)scalce_patterns_buckets(ctx) + 1;
    memset(res, 0, sizeof *res);
    if (reuse) {
      res->counts = keep.counts;
      res->name_bytes = keep.name_bytes;
      memcpy(res->keep, keep.keep, sizeof keep.keep);
      memcpy(res->keep_bytes, keep.keep_bytes, sizeof keep.keep_bytes);
    }
    res->magic = 0x5CA1CE5Du;
  }
  res->world = W;
  res->rank = rank;
  // The reordered q' stream of this call is only passed on (step 8): it lives in the workspace the batch shares with the other
  // shards in flight, not in the batch -- 5 GB less per slot at 50 M reads of 100 bp.  (Not under -A: the stream is the output.)
  struct StreamScratch {
    scalce_batch *b; bool on;
    ~StreamScratch() { if (on) scalce_batch_set_stream_scratch(b, 0); }
  } stream_scratch{b, scalce_batch_set_stream_scratch(b, 1) == SCALCE_OK};
  static thread_local std::string last_error;
  // SCALCE_TRACE=1: where a rank's time goes (the stream is drained at every mark: for looking, not for timing runs)
  const bool trace = getenv("SCALCE_TRACE") != nullptr;
  double t_last = 0;
  auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
  auto mark = [&](const char *what) {
    if (!trace) return;
    hipStreamSynchronize(s);
    const double t = now();
    if (t_last > 0) fprintf(stderr, "  [rank %d] %-28s %8.2f ms\n", rank, what, (t - t_last) * 1e3);
    t_last = t;
  };
  mark("start");
  try {
    DevMem mem;
    const uint8_t *text[2] = {d_text1, d_text2};
    u64 nbytes[2] = {n1, n2};
    const void *dp = nullptr;
    uint64_t nb = 0;
    // ---- 1. first pass over the own piece: rows + quality counters, no tie-break yet
    // A failure that only ONE rank sees -- a malformed record in its byte range, a read of another length, a buffer that
    // is too small -- must not leave the others waiting in the next collective for ever.  Rank-local work runs under
    // local(): what it throws is kept, the rank goes on to the next exchange with empty hands, and the status travels with
    // an exchange that happens anyway (or with agree(), one word per rank) -- then every rank throws together.
    Fail local_err{"", 0};
    auto local = [&](auto &&fn) {
      if (local_err.rc) return;
      try { fn(); } catch (const Fail &f) { local_err = f; if (!local_err.rc) local_err.rc = SCALCE_ERR_HIP; }
    };
    auto together = [&](const std::vector<u64> &status, size_t stride, size_t at, const char *where) {
      for (int r = 0; r < W; r++)
        if (status[(size_t)r * stride + at]) {
          if (r == rank && local_err.rc) throw local_err;
          throw Fail{std::string("rank ") + std::to_string(r) + " failed (" + where + "); see its message", (int)status[(size_t)r * stride + at]};
        }
      if (local_err.rc) throw local_err;  // (cannot happen: the own status was in the exchange)
    };
    uint64_t used[2] = {0, 0};
    u64 N0 = 0;
    const int nm = text[1] ? 2 : 1;
    int L[2] = {0, 0};
    local([&] {
      SH_RC(ctx, scalce_batch_reset(b));
      SH_RC(ctx, scalce_batch_append(b, text[0], nbytes[0], text[1], nbytes[1], SCALCE_APPEND_FINAL | SCALCE_APPEND_NO_TOKENIZE, used, s));
      N0 = scalce_batch_reads(b);
      for (int m = 0; m < nm; m++) {  // read lengths: symbols per row
        uint8_t e4[4];
        uint64_t ns = 0;
        int32_t rl = 0;
        SH_RC(ctx, scalce_batch_qinput_edges(b, m, e4, &ns, &rl, s));
        L[m] = N0 ? rl : 0;
      }
    });
    if (local_err.rc) N0 = 0;
    mark("first pass (ingest+quality)");
    const uint32_t nb1 = (uint32_t)scalce_patterns_buckets(ctx) + 1;
    res->nb1 = nb1;
    u64 *d_small = mem.alloc<u64>(4096 + 64 * (size_t)W);
    u64 *d_gather = mem.alloc<u64>((size_t)W * 4096 + (size_t)W * 64 * W);
    auto agree = [&](const char *where) {  // one status word per rank
      u64 mine = (u64)local_err.rc;
      std::vector<u64> all = gather_host<u64>(comm, &mine, 1, d_small, d_gather, s);
      together(all, 1, 0, where);
    };
    // rows and read lengths of everyone (a rank without reads learns the read length here), and how the first pass went
    std::vector<u64> meta;
    {
      u64 mine[4] = {N0, (u64)L[0], (u64)L[1], (u64)local_err.rc};
      meta = gather_host<u64>(comm, mine, 4, d_small, d_gather, s);
      together(meta, 4, 3, "first pass over its piece of the input");
      for (int r = 0; r < W; r++)
        for (int m = 0; m < nm; m++) {
          const int Lr = (int)meta[4 * r + 1 + m];
          if (!L[m] && Lr) L[m] = Lr;
          // every rank sees the same table: they all stop here together (the reference: compress.cpp:628-634)
          if (Lr && L[m] && Lr != L[m]) throw Fail{"(ERROR) reads of different lengths in the input (" + std::to_string(L[m]) + " vs " + std::to_string(Lr) + ")", SCALCE_ERR_FORMAT};
        }
    }
    std::vector<u64> g(W + 1, 0);  // run-wide index of every rank's first record
    for (int r = 0; r < W; r++) g[r + 1] = g[r] + meta[4 * r];
    const u64 total_reads = g[W];
    res->reads_total = total_reads;
    // the first two and last two q' symbols of the own piece: trigrams that straddle a rank boundary (step 4)
    uint8_t edge[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int m = 0; m < nm; m++) {
      uint64_t ns = 0;
      SH_RC(ctx, scalce_batch_qinput_edges(b, m, edge[m], &ns, nullptr, s));
      nb = ns;
    }
    // ---- 2. spill chunks of the run-wide -B rule: a chain of carries, then everybody learns every cut
    std::vector<u64> cuts_global;  // run-wide rows in front of which a chunk begins
    {
      const uint32_t CAP = 4000;
      std::vector<uint64_t> cuts(CAP);
      uint32_t nc = 0;
      uint64_t carry_out = 0;
      local([&] { SH_RC(ctx, scalce_batch_chunk_plan(b, 0, cuts.data(), CAP, &nc, &carry_out, s)); });  // sizes of all rows: every rank at once
      u64 carry_in = 0;
      for (int k = 0; k < W; k++) {  // rank k can cut once it knows what rank k - 1 left open
        if (k == rank) local([&] { SH_RC(ctx, scalce_batch_chunk_plan(b, carry_in, cuts.data(), CAP, &nc, &carry_out, s)); });
        if (W > 1) {
          u64 mine = k == rank ? carry_out : 0;
          std::vector<u64> all = gather_host<u64>(comm, &mine, 1, d_small, d_gather, s);
          if (rank == k + 1) carry_in = all[k];
        }
      }
      local([&] { if (nc >= CAP) throw Fail{"more than 4000 spill chunks on one rank: -B is too small for this input", SCALCE_ERR_CAPACITY}; });
      if (local_err.rc) nc = 0;
      std::vector<u64> mine(CAP + 2, 0);
      mine[0] = nc;
      mine[CAP + 1] = (u64)local_err.rc;
      for (uint32_t i = 0; i < nc; i++) mine[1 + i] = g[rank] + cuts[i];
      std::vector<u64> all = gather_host<u64>(comm, mine.data(), CAP + 2, d_small, d_gather, s);
      together(all, CAP + 2, CAP + 1, "spill-chunk plan");
      for (int r = 0; r < W; r++)
        for (u64 i = 0; i < all[(size_t)r * (CAP + 2)]; i++) cuts_global.push_back(all[(size_t)r * (CAP + 2) + 1 + i]);
      std::sort(cuts_global.begin(), cuts_global.end());
      res->chunks_total = (uint32_t)cuts_global.size() + ((cuts_global.empty() || cuts_global.back() < total_reads) ? 1 : 0);
    }
    mark("record sizes + cuts");
    // ---- 3. rank boundaries move to the nearest cut; the records in between change owner as text
    std::vector<u64> gn(W + 1);
    {
      const int prc = scalce_shard_plan_boundaries(W, g.data(), cuts_global.data(), cuts_global.size(), gn.data());
      // (every rank computes the same plan from the same cuts: they all leave here together)
      if (prc) throw Fail{"internal: rank boundaries could not be planned", SCALCE_ERR_ARG};
    }
    auto clampu = [](u64 x, u64 a, u64 b) { return x < a ? a : (x > b ? b : x); };
    res->first_read = gn[rank];
    res->reads_local = gn[rank + 1] - gn[rank];
    res->moved_in[0] = gn[rank] < g[rank] ? g[rank] - clampu(gn[rank], 0, g[rank]) : 0;
    res->moved_in[1] = gn[rank + 1] > g[rank + 1] ? gn[rank + 1] - g[rank + 1] : 0;
    bool anything_moves = false;
    for (int r = 1; r < W; r++) anything_moves = anything_moves || gn[r] != g[r];
    if (anything_moves) {
      // Rows [g[rank], g[rank + 1]) were ingested here; rows [gn[d], gn[d + 1]) belong to rank d now.  My rows go to their new
      // owners in rank order, which is the order they have in my text: the text itself is the send buffer, every range where it
      // lies.  What arrives from the ranks in front goes IN FRONT of the rows that stay, what arrives from the ranks behind goes
      // behind them; the rows that stay are not ingested again (scalce_batch_rewindow; rounds 1-4 rebuilt the whole range from
      // text: a second ingest and a second first walk of every row).
      const uint8_t *front_text[2] = {nullptr, nullptr}, *back_text[2] = {nullptr, nullptr};
      uint64_t front_bytes[2] = {0, 0}, back_bytes[2] = {0, 0};
      const u64 keep_lo = clampu(gn[rank], g[rank], g[rank + 1]), keep_hi = clampu(gn[rank + 1], g[rank], g[rank + 1]);
      for (int m = 0; m < nm; m++) {
        std::vector<uint64_t> sendb(W, 0), recvb(W, 0), start(W + 1, 0);
        for (int d = 0; d <= W; d++) {
          const u64 row = clampu(d < W ? gn[d] : gn[W], g[rank], g[rank + 1]) - g[rank];
          uint64_t off = nbytes[m];
          if (row < N0) local([&] { SH_RC(ctx, scalce_batch_text_offset(b, m, row, &off, s)); });
          start[d] = off;
        }
        start[0] = 0;  // (rows in front of gn[0] = 0 do not exist)
        for (int d = 0; d < W; d++) sendb[d] = start[d + 1] - start[d];
        // who receives how much from whom
        std::vector<u64> all = gather_host<u64>(comm, sendb.data(), (size_t)W, d_small, d_gather, s);
        for (int src = 0; src < W; src++) recvb[src] = all[(size_t)src * W + rank];
        u64 before_self = 0, after_self = 0;
        for (int src = 0; src < W; src++) { if (src < rank) before_self += recvb[src]; else if (src > rank) after_self += recvb[src]; }
        // everything but the rows that stay goes through the transport; the two parts land 16-byte aligned (the ingest kernels
        // read their text in aligned 16-byte words)
        const u64 back_at = (before_self + 15) & ~15ull;
        uint8_t *d_recv = mem.alloc<uint8_t>(back_at + after_self + 256);
        std::vector<uint64_t> sb = sendb, rb = recvb, so(W), ro(W);
        sb[rank] = 0;
        rb[rank] = 0;
        u64 fa = 0, ba = back_at;
        for (int r = 0; r < W; r++) {
          so[r] = start[r];
          if (r < rank) { ro[r] = fa; fa += rb[r]; } else { ro[r] = ba; ba += rb[r]; }
        }
        SH_CM(comm, scalce_comm_all_to_all_vo(comm, text[m], so.data(), sb.data(), d_recv, ro.data(), rb.data(), s));
        front_text[m] = d_recv; front_bytes[m] = before_self;
        back_text[m] = d_recv + back_at; back_bytes[m] = after_self;
      }
      // the rows of the new range (their quality symbols were counted by whoever held them in the first pass)
      local([&] {
        SH_RC(ctx, scalce_batch_rewindow(b, keep_lo - g[rank], keep_hi - keep_lo, front_text, front_bytes, back_text, back_bytes, s));
        if (scalce_batch_reads(b) != res->reads_local) throw Fail{"internal: row count after the exchange differs from the plan", SCALCE_ERR_ARG};
      });
    }
    if (W > 1) agree("exchange of rows between ranks");
    else if (local_err.rc) throw local_err;
    mark("exchange + re-ingest");
    const u64 N = scalce_batch_reads(b);
    // ---- 4. run-wide quality model: trigrams across the ORIGINAL piece boundaries, all-reduce, scaling
    uint32_t *d_table = nullptr;
    scalce_params bp;
    SH_RC(ctx, scalce_batch_params(b, &bp));
    if (L[0] && !bp.no_ac) {  // (-A: no statistics, the q' rows are stored as they are)
      std::vector<uint8_t> ed(8, 0);
      memcpy(ed.data(), edge, 8);
      uint8_t *d_e = reinterpret_cast<uint8_t *>(d_small);
      std::vector<uint8_t> edges = gather_host<uint8_t>(comm, ed.data(), 8, d_e, reinterpret_cast<uint8_t *>(d_gather), s);
      d_table = keep_alloc<uint32_t>(res, 2, 2 * 512000);
      for (int m = 0; m < nm; m++) {
        SH_RC(ctx, scalce_batch_output(b, SCALCE_OUT_FREQ4, m, &dp, &nb));
        u64 *d_f4 = static_cast<u64 *>(const_cast<void *>(dp));
        // symbols next to the boundary in front of this rank's ORIGINAL piece (pieces before it may be empty or short)
        std::vector<uint32_t> keys;
        const u64 own = meta[4 * rank] * (u64)L[m];
        if (rank > 0 && own) {
          std::vector<uint32_t> before;  // up to two symbols in front of the piece, nearest last
          for (int r = rank - 1; r >= 0 && before.size() < 2; r--) {
            const u64 nsy = meta[4 * r] * (u64)L[m];
            const uint8_t *e = &edges[(size_t)r * 8 + 4 * m];
            if (nsy >= 2) { if (before.empty()) { before.push_back(e[3]); before.push_back(e[2]); } else before.push_back(e[3]); }
            else if (nsy == 1) before.push_back(e[0]);
          }
          const uint8_t *me = &edges[(size_t)rank * 8 + 4 * m];
          const uint32_t s0 = me[0], s1 = own >= 2 ? me[1] : 0xFFFFu;
          // before[0] = the symbol right in front (p1), before[1] = the one before that (p0)
          if (before.size() >= 2 && s0 < 80 && before[0] < 80 && before[1] < 80) keys.push_back((before[1] * 80 + before[0]) * 80 + s0);
          if (before.size() >= 1 && own >= 2 && s0 < 80 && s1 < 80 && before[0] < 80) keys.push_back((before[0] * 80 + s0) * 80 + s1);
        }
        if (!keys.empty()) {
          uint32_t *d_k = reinterpret_cast<uint32_t *>(d_small);
          SH_HIP(hipMemcpyAsync(d_k, keys.data(), keys.size() * 4, hipMemcpyHostToDevice, s));
          hipLaunchKernelGGL(add_ones_k, dim3(1), dim3(1), 0, s, d_f4, d_k, (uint32_t)keys.size());
          SH_HIP(hipStreamSynchronize(s));
        }
        SH_CM(comm, scalce_comm_all_reduce_sum_u64(comm, reinterpret_cast<uint64_t *>(d_f4), 512000, s));
        const u64 symbols = total_reads * (u64)L[m];
        SH_RC(ctx, scalce_ac_scale(ctx, reinterpret_cast<const uint64_t *>(d_f4), (uint32_t)(1 + symbols / 0xFFFFFFFFull), d_table + 512000 * (size_t)m, s));  // compress.cpp:297-303
      }
    }
    mark("quality model");
    // ---- 5. tie-break across ranks.  bin_size is cumulative over the run (reads.cpp:246): a rank's tie reads are decided
    // against the FINAL counts of every rank in front of it.  Rounds 1-3 iterated: all-gather of everybody's counts, a few
    // global sweeps, until no rank moved (~13 rounds, 27 ms per 50 M-read shard).  Round 4: the ranks form a chain.  Rank r
    // waits for the counts of ranks 0 .. r-1 from rank r-1 (one message: [status][buckets + 1 counts]), settles its own tie
    // reads against them window by window (scalce_batch_tokenize_settle: the tie-break of a batch on its own, exact given its
    // prior), adds its counts and hands the sum to rank r+1.  A rank only spends its own settle; what the chain adds is a
    // start-up skew of one settle per rank in front -- with several shards in flight the ranks work on different shards.
    {
      const uint32_t stride = nb1 + 1;
      u64 *d_msg = nullptr;  // the chain's message: [0] status of the ranks in front (0 = fine), [1 ..] their counts per bucket
      local([&] { d_msg = mem.alloc<u64>(stride); });  // (before the agree(): a rank that cannot even hold the message stops everybody there)
      local([&] { SH_RC(ctx, scalce_batch_tokenize_begin(b, s)); });
      if (W > 1) agree("tokenizer"); else if (local_err.rc) throw local_err;
      {
        // Everything between the receive and the send runs under local(): whatever fails here -- the receive itself, a copy,
        // the settle, the counts -- this rank still SENDS (status = its error code), so the rank behind never sits in a receive
        // that nobody answers (RCCL has no timeout), and every rank reaches the agree() below and throws there together.
        u64 upstream = 0;
        if (rank > 0) {
          // (the receive is posted even when this rank has already failed: the rank in front sends in any case, and a send
          //  that nobody takes would block its stream in front of the agree() below)
          try {
            SH_CM(comm, scalce_comm_recv(comm, d_msg, (size_t)stride * 8, rank - 1, s));
            SH_HIP(hipMemcpyAsync(&upstream, d_msg, 8, hipMemcpyDeviceToHost, s));
            SH_HIP(hipStreamSynchronize(s));
          } catch (const Fail &f) {
            if (!local_err.rc) { local_err = f; if (!local_err.rc) local_err.rc = SCALCE_ERR_HIP; }
          }
        } else {
          local([&] { SH_HIP(hipMemsetAsync(d_msg, 0, (size_t)stride * 8, s)); });
        }
        if (!upstream)
          local([&] { SH_RC(ctx, scalce_batch_tokenize_settle(b, rank > 0 ? reinterpret_cast<const uint64_t *>(d_msg + 1) : nullptr, s)); });
        else
          local([&] { SH_RC(ctx, scalce_batch_tokenize_settle(b, nullptr, s)); });  // (a rank in front failed: the run ends at the next agree())
        res->rounds = 1;
        res->sweeps = 0;
        if (rank + 1 < W) {
          local([&] {
            SH_RC(ctx, scalce_batch_output(b, SCALCE_OUT_BUCKET_COUNTS, 0, &dp, &nb));
            hipLaunchKernelGGL(add_counts_into_k, dim3((nb1 + 255) / 256), dim3(256), 0, s, nb1, static_cast<const u64 *>(dp), d_msg + 1);
          });
          // the message goes out whatever happened above; only a send that itself fails is left to the transport's own error
          // (the peer of a dead communicator gets an error from its receive, not a hang)
          const u64 status = upstream ? upstream : (local_err.rc ? 2 + (u64)local_err.rc : 0);
          hipError_t he = hipMemcpyAsync(d_msg, &status, 8, hipMemcpyHostToDevice, s);
          if (he == hipSuccess) he = hipStreamSynchronize(s);  // `status` is a stack variable
          const int src = scalce_comm_send(comm, d_msg, (size_t)stride * 8, rank + 1, s);
          if (!local_err.rc) {
            if (he != hipSuccess) local_err = Fail{std::string("tie-break chain, status word: ") + hipGetErrorString(he), SCALCE_ERR_HIP};
            else if (src) local_err = Fail{std::string("scalce_comm_send (tie-break chain): ") + scalce_comm_error(comm), src};
          }
        }
      }
    }
    mark("tie-break");
    // ---- 6. order (the run's cuts inside this rank's rows are its chunks) and emit
    {
      std::vector<uint64_t> starts(1, 0);
      for (u64 c : cuts_global)
        if (c > gn[rank] && c < gn[rank + 1]) starts.push_back(c - gn[rank]);
      local([&] {
        SH_RC(ctx, scalce_batch_set_chunks(b, starts.data(), (uint32_t)starts.size()));
        SH_RC(ctx, scalce_batch_order(b, s));
        SH_RC(ctx, scalce_batch_emit(b, s));
        SH_RC(ctx, scalce_batch_set_chunks(b, nullptr, 0));
      });
      if (W > 1) agree("order and emit stages"); else if (local_err.rc) throw local_err;
    }
    mark("order + emit");
    // ---- 7. who holds how much of every bucket
    if (!res->counts) res->counts = static_cast<uint64_t *>(calloc((size_t)W * nb1, 8));
    if (!res->name_bytes) res->name_bytes = static_cast<uint64_t *>(calloc((size_t)W * nb1, 8));
    memset(res->name_bytes, 0, (size_t)W * nb1 * 8);
    {
      u64 *d_all = mem.alloc<u64>((size_t)W * nb1);
      SH_RC(ctx, scalce_batch_output(b, SCALCE_OUT_BUCKET_COUNTS, 0, &dp, &nb));
      SH_CM(comm, scalce_comm_all_gather(comm, dp, d_all, (size_t)nb1 * 8, s));
      SH_HIP(hipMemcpyAsync(res->counts, d_all, (size_t)W * nb1 * 8, hipMemcpyDeviceToHost, s));
      SH_HIP(hipStreamSynchronize(s));
      SH_RC(ctx, scalce_batch_output(b, SCALCE_OUT_BUCKET_NAME_BYTES, 0, &dp, &nb));
      if (nb) {
        SH_CM(comm, scalce_comm_all_gather(comm, dp, d_all, (size_t)nb1 * 8, s));
        SH_HIP(hipMemcpyAsync(res->name_bytes, d_all, (size_t)W * nb1 * 8, hipMemcpyDeviceToHost, s));
        SH_HIP(hipStreamSynchronize(s));
      }
    }
    mark("bucket layout");
    // ---- 8. the run-wide reordered quality stream in contiguous block ranges, one range per rank; code it
    if (d_table && L[0]) {
      const uint64_t *C = res->counts;
      for (int m = 0; m < nm; m++) {
        const u64 Lm = (u64)L[m];
        std::vector<uint64_t> sendb(W), recvb(W), psrc((size_t)W * nb1), pdst((size_t)W * nb1);
        uint64_t lo = 0, hi = 0, np = 0;
        if (scalce_shard_plan_blocks(W, rank, nb1, C, Lm, sendb.data(), recvb.data(), &lo, &hi, psrc.data(), pdst.data(), &np))
          throw Fail{"internal: block-range plan is inconsistent", SCALCE_ERR_ARG};
        psrc.resize(np);
        pdst.resize(np);
        u64 recv_total = 0;
        for (int r = 0; r < W; r++) recv_total += recvb[r];
        SH_RC(ctx, scalce_batch_output(b, SCALCE_OUT_QSTREAM, m, &dp, &nb));
        if (nb != N * Lm) throw Fail{"internal: reordered stream has an unexpected size", SCALCE_ERR_ARG};
        // What stays on this rank never goes through the transport: its pieces are copied straight from the local stream
        // (at one rank that is everything: 5 GB per 50 M-read shard that round 4 sent to itself through ncclSend / ncclRecv,
        // 16 of that path's 19 ms).  The receive buffer only holds what other ranks send, in rank order.
        u64 self_at = 0, local_off = 0;   // where the own bytes would lie in a full receive buffer / where they lie in the local stream
        for (int r = 0; r < rank; r++) { self_at += recvb[r]; local_off += sendb[r]; }
        const u64 self_bytes = recvb[rank];
        if (self_bytes != sendb[rank]) throw Fail{"internal: block-range plan disagrees with itself about the bytes that stay", SCALCE_ERR_ARG};
        std::vector<uint64_t> sb = sendb, rb = recvb;
        sb[rank] = 0;
        rb[rank] = 0;
        uint8_t *d_got = mem.alloc<uint8_t>(recv_total - self_bytes);
        uint8_t *d_mine = keep_alloc<uint8_t>(res, m, hi - lo + 16);
        // (the send buffer is the local stream itself, every range where it lies; the own range is the hole in the middle)
        if (W > 1) {
          std::vector<uint64_t> so(W), ro(W);
          u64 a = 0, g = 0;
          for (int r = 0; r < W; r++) { so[r] = a; ro[r] = g; a += sendb[r]; g += rb[r]; }
          SH_CM(comm, scalce_comm_all_to_all_vo(comm, dp, so.data(), sb.data(), d_got, ro.data(), rb.data(), s));
        }
        if (!psrc.empty()) {
          // pieces are source-major: [from ranks in front | own | from ranks behind]; the own ones read the local stream
          std::vector<uint64_t> ps_o, pd_o, ps_x, pd_x;
          u64 own_total = 0, other_total = 0;
          for (size_t i = 0; i < psrc.size(); i++) {
            const u64 len = (i + 1 < psrc.size() ? psrc[i + 1] : recv_total) - psrc[i];
            if (psrc[i] >= self_at && psrc[i] < self_at + self_bytes) { ps_o.push_back(psrc[i] - self_at); pd_o.push_back(pdst[i]); own_total += len; }
            else { ps_x.push_back(psrc[i] < self_at ? psrc[i] : psrc[i] - self_bytes); pd_x.push_back(pdst[i]); other_total += len; }
          }
          if (own_total != self_bytes || other_total != recv_total - self_bytes) throw Fail{"internal: block-range pieces do not add up", SCALCE_ERR_ARG};
          uint64_t *d_ps = mem.alloc<uint64_t>(psrc.size()), *d_pd = mem.alloc<uint64_t>(psrc.size());
          if (!ps_x.empty()) {
            SH_HIP(hipMemcpyAsync(d_ps, ps_x.data(), ps_x.size() * 8, hipMemcpyHostToDevice, s));
            SH_HIP(hipMemcpyAsync(d_pd, pd_x.data(), pd_x.size() * 8, hipMemcpyHostToDevice, s));
            SH_RC(ctx, scalce_copy_pieces(ctx, d_got, d_mine, d_ps, d_pd, (uint32_t)ps_x.size(), other_total, s));
          }
          if (!ps_o.empty()) {
            uint64_t *d_ps2 = d_ps + ps_x.size(), *d_pd2 = d_pd + ps_x.size();
            SH_HIP(hipMemcpyAsync(d_ps2, ps_o.data(), ps_o.size() * 8, hipMemcpyHostToDevice, s));
            SH_HIP(hipMemcpyAsync(d_pd2, pd_o.data(), pd_o.size() * 8, hipMemcpyHostToDevice, s));
            SH_RC(ctx, scalce_copy_pieces(ctx, static_cast<const uint8_t *>(dp) + local_off, d_mine, d_ps2, d_pd2, (uint32_t)ps_o.size(), own_total, s));
          }
          SH_HIP(hipStreamSynchronize(s));  // the piece lists are host vectors of this scope
        }
        res->sym_lo[m] = lo;
        res->sym_hi[m] = hi;
        const uint32_t *tab = d_table + 512000 * (size_t)m;
        if (flags & SCALCE_SHARD_PREPARE_ONLY) SH_RC(ctx, scalce_batch_entropy_stream_prepare(b, m, tab, d_mine, hi - lo, s));
        else if (flags & SCALCE_SHARD_CODER_ASYNC) {
          hipEvent_t ev;
          SH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
          SH_HIP(hipEventRecord(ev, s));
          SH_HIP(hipStreamWaitEvent((hipStream_t)coder_stream, ev, 0));
          SH_HIP(hipEventDestroy(ev));
          SH_RC(ctx, scalce_batch_entropy_stream_begin(b, m, tab, d_mine, hi - lo, coder_stream));
        } else {
          SH_RC(ctx, scalce_batch_entropy_stream(b, m, tab, d_mine, hi - lo, s));
        }
      }
      if (!(flags & (SCALCE_SHARD_PREPARE_ONLY | SCALCE_SHARD_CODER_ASYNC))) {
        SH_RC(ctx, scalce_batch_finish(b, s));
        u64 mine[2] = {0, 0};
        for (int m = 0; m < nm; m++) { SH_RC(ctx, scalce_batch_output(b, SCALCE_OUT_QUAL, m, &dp, &nb)); mine[m] = nb; }
        std::vector<u64> all = gather_host<u64>(comm, mine, 2, d_small, d_gather, s);
        for (int r = 0; r < W; r++)
          for (int m = 0; m < 2; m++) res->coded_bytes[m][r] = all[2 * r + m];
      }
    }
    mark("block ranges + coder");
    SH_HIP(hipStreamSynchronize(s));
    return SCALCE_OK;
  } catch (const Fail &f) {
    last_error = f.msg;
    fprintf(stderr, "scalce_sharded_compress (rank %d of %d): %s\n", rank, W, f.msg.c_str());
    scalce_shard_result_free(res);
    return f.rc ? f.rc : SCALCE_ERR_HIP;
  }
}
