// stream_host.cpp -- streaming host of the compress path: files of any size through one GPU.
//
// What it replaces in the reference: the reader half of thread() (compress.cpp:614-671: records read one by one under
// r_spin from zlib-backed files) together with the reason for the spill files (compress.cpp:708-715: the bucket pool is
// bounded).  Here the TEXT is what is bounded: reader threads (one per mate) fill pinned chunks, the chunks go up with
// hipMemcpyAsync while the previous piece is ingested / counted / tokenized (scalce_batch_append), and only the rows
// derived from the text stay in HBM.  Order, emit and entropy then run once over the whole run, so the archive is the one
// a single resident shard gives -- and -B chunks are cut on run-wide record sizes exactly like the reference's.
//
// Built on the public C ABI only (include/scalce_hip.h) plus the HIP runtime for pinned memory and copies.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/scalce_hip.h"

namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// pinned chunks of one mate's stream: filled by the reader thread, drained by the upload
struct ChunkRing {
  struct Chunk {
    uint8_t *p = nullptr;
    uint64_t n = 0;
    bool last = false;  // nothing behind this chunk
  };
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Chunk> free_, full_;
  std::string error;
  bool failed = false, stop = false;

  Chunk take_free() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return !free_.empty() || stop; });
    if (stop) return Chunk();
    Chunk c = free_.front();
    free_.pop_front();
    return c;
  }
  void put_full(const Chunk &c) {
    { std::lock_guard<std::mutex> lk(mu); full_.push_back(c); }
    cv.notify_all();
  }
  bool take_full(Chunk &c) {  // false: the reader failed
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return !full_.empty() || failed; });
    if (full_.empty()) return false;
    c = full_.front();
    full_.pop_front();
    return true;
  }
  void put_free(const Chunk &c) {
    { std::lock_guard<std::mutex> lk(mu); free_.push_back(c); }
    cv.notify_all();
  }
  void fail(const std::string &msg) {
    { std::lock_guard<std::mutex> lk(mu); failed = true; error = msg; }
    cv.notify_all();
  }
  void shutdown() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
  }
};

void reader_main(ChunkRing *ring, scalce_read_fn rd, void *user, uint64_t piece) {
  for (;;) {
    ChunkRing::Chunk c = ring->take_free();
    if (!c.p) return;
    c.n = 0;
    c.last = false;
    while (c.n < piece) {
      const int64_t k = rd(user, c.p + c.n, piece - c.n);
      if (k < 0) { ring->fail("read error on the input stream"); return; }
      if (k == 0) { c.last = true; break; }
      c.n += (uint64_t)k;
    }
    ring->put_full(c);
    if (c.last) return;
  }
}

struct Mate {
  ChunkRing ring;
  std::thread reader;
  std::vector<uint8_t *> pinned;
  // the piece handed to the batch: [tail of the previous piece][new chunks]; two buffers, so that the tail moves to the
  // front of the OTHER one (no overlapping copy)
  uint8_t *d_text[2] = {nullptr, nullptr};
  int cur = 0;
  uint64_t have = 0;              // bytes of d_text[cur] in use
  // chunks on their way up: a queue of at most two, slot = index & 1
  uint8_t *d_raw[2] = {nullptr, nullptr};
  uint64_t raw_n[2] = {0, 0};
  hipEvent_t up_ev[2] = {nullptr, nullptr};    // the chunk has arrived in d_raw[slot]
  hipEvent_t land_ev[2] = {nullptr, nullptr};  // ... and has been copied behind the tail: the slot may be overwritten
  bool landed_once[2] = {false, false};
  ChunkRing::Chunk raw_chunk[2];
  uint64_t head = 0, tail = 0;    // queue of chunks in d_raw: [head, tail)
  bool eof = false;               // the stream's last chunk has been taken from the ring
  bool all_landed() const { return eof && head == tail; }
};

}  // namespace

#define ST_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); goto fail; } \
  } while (0)

extern "C" int scalce_stream_compress(scalce_ctx *ctx, const scalce_params *p, scalce_read_fn rd1, void *user1, scalce_read_fn rd2,
                                      void *user2, uint64_t piece_bytes, uint64_t reads_hint, int flags, scalce_batch **out,
                                      scalce_stream_stats *st, char *errbuf, size_t errcap) {
  if (!ctx || !p || !rd1 || !out || (p->paired && !rd2)) return SCALCE_ERR_ARG;
  const int nm = p->paired ? 2 : 1;
  const uint64_t piece = ((piece_bytes ? piece_bytes : (1ull << 30)) + 4095) & ~4095ull;
  const int NPIN = 3;
  const uint64_t cap = 2 * piece;  // of a piece: the tail of the previous one plus one chunk
  scalce_read_fn rd[2] = {rd1, rd2};
  void *user[2] = {user1, user2};
  Mate M[2];
  scalce_batch *b = nullptr;
  hipStream_t s_copy = nullptr, s_main = nullptr;
  std::string err;
  int rc = SCALCE_OK;
  scalce_stream_stats S;
  memset(&S, 0, sizeof S);
  const double t_start = now_s();
  bool started = false;

  ST_HIP(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
  ST_HIP(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
  for (int m = 0; m < nm; m++) {
    for (int i = 0; i < NPIN; i++) {
      void *hp = nullptr;
      ST_HIP(hipHostMalloc(&hp, piece, hipHostMallocDefault));
      M[m].pinned.push_back(static_cast<uint8_t *>(hp));
      ChunkRing::Chunk c;
      c.p = static_cast<uint8_t *>(hp);
      M[m].ring.free_.push_back(c);
    }
    for (int i = 0; i < 2; i++) {
      ST_HIP(hipMalloc(reinterpret_cast<void **>(&M[m].d_text[i]), cap + 256));
      ST_HIP(hipMalloc(reinterpret_cast<void **>(&M[m].d_raw[i]), piece + 256));
      ST_HIP(hipEventCreateWithFlags(&M[m].up_ev[i], hipEventDisableTiming));
      ST_HIP(hipEventCreateWithFlags(&M[m].land_ev[i], hipEventDisableTiming));
    }
  }
  {
    const uint64_t rows0 = reads_hint ? reads_hint + 16 : piece / (2 * (uint64_t)p->read_len[0] + 7) + 16;
    rc = scalce_batch_create(ctx, p, rows0, cap + 256, &b);
    if (rc) { err = scalce_last_error(ctx); goto fail_rc; }
    scalce_batch_set_lean(b, (flags & SCALCE_STREAM_LEAN) ? 1 : 0);
  }
  for (int m = 0; m < nm; m++) M[m].reader = std::thread(reader_main, &M[m].ring, rd[m], user[m], piece);
  started = true;

  {
    // one more chunk of every mate that has input left and a free slot starts its way up
    auto upload = [&]() -> int {
      for (int m = 0; m < nm; m++) {
        Mate &x = M[m];
        if (x.eof || x.tail - x.head >= 2) continue;
        const int sl = (int)(x.tail & 1);
        ChunkRing::Chunk c;
        const double t0 = now_s();
        if (!x.ring.take_full(c)) { err = x.ring.error; return SCALCE_ERR_FORMAT; }
        S.read_wait_s += now_s() - t0;
        if (c.last) x.eof = true;
        x.raw_chunk[sl] = c;
        x.raw_n[sl] = c.n;
        // the slot's previous chunk must have left for d_text before this one lands on it
        if (x.landed_once[sl] && hipStreamWaitEvent(s_copy, x.land_ev[sl], 0) != hipSuccess) { err = "hipStreamWaitEvent failed"; return SCALCE_ERR_HIP; }
        if (c.n && hipMemcpyAsync(x.d_raw[sl], c.p, c.n, hipMemcpyHostToDevice, s_copy) != hipSuccess) { err = "hipMemcpyAsync (chunk upload) failed"; return SCALCE_ERR_HIP; }
        if (hipEventRecord(x.up_ev[sl], s_copy) != hipSuccess) { err = "hipEventRecord failed"; return SCALCE_ERR_HIP; }
        x.tail++;
        S.bytes[m] += c.n;
      }
      return SCALCE_OK;
    };
    // chunks that have arrived go behind the tail in d_text while the piece is short of `piece` bytes and they fit;
    // their pinned buffers return to the reader
    int landed = 0;
    auto land = [&]() -> int {
      landed = 0;
      for (int m = 0; m < nm; m++) {
        Mate &x = M[m];
        while (x.head < x.tail && x.have < piece && x.have + x.raw_n[x.head & 1] <= cap) {
          const int sl = (int)(x.head & 1);
          const double t0 = now_s();
          if (hipEventSynchronize(x.up_ev[sl]) != hipSuccess) { err = "chunk upload failed"; return SCALCE_ERR_HIP; }
          S.h2d_wait_s += now_s() - t0;
          if (x.raw_n[sl] && hipMemcpyAsync(x.d_text[x.cur] + x.have, x.d_raw[sl], x.raw_n[sl], hipMemcpyDeviceToDevice, s_main) != hipSuccess) {
            err = "hipMemcpyAsync (piece assembly) failed";
            return SCALCE_ERR_HIP;
          }
          if (hipEventRecord(x.land_ev[sl], s_main) != hipSuccess) { err = "hipEventRecord failed"; return SCALCE_ERR_HIP; }
          x.landed_once[sl] = true;
          x.have += x.raw_n[sl];
          x.ring.put_free(x.raw_chunk[sl]);
          x.head++;
          landed++;
        }
      }
      return SCALCE_OK;
    };

    if ((rc = upload())) goto fail_rc;
    for (;;) {
      if ((rc = land())) goto fail_rc;
      if ((rc = upload())) goto fail_rc;  // travels while this piece is worked on
      bool final_piece = true;
      for (int m = 0; m < nm; m++) final_piece = final_piece && M[m].all_landed();
      uint64_t used[2] = {0, 0};
      const double t0 = now_s();
      rc = scalce_batch_append(b, M[0].d_text[M[0].cur], M[0].have, nm == 2 ? M[1].d_text[M[1].cur] : nullptr, nm == 2 ? M[1].have : 0,
                               final_piece ? 1 : 0, used, s_main);
      S.front_s += now_s() - t0;
      if (rc) { err = scalce_last_error(ctx); goto fail_rc; }
      S.rounds++;
      bool progress = landed > 0;
      for (int m = 0; m < nm; m++) {
        Mate &x = M[m];
        if (used[m] > x.have) { err = "internal: consumed more than the piece"; rc = SCALCE_ERR_ARG; goto fail_rc; }
        progress = progress || used[m] > 0;
        const uint64_t rest = x.have - used[m];
        if (rest && used[m]) {  // the unconsumed tail opens the next piece
          ST_HIP(hipMemcpyAsync(x.d_text[x.cur ^ 1], x.d_text[x.cur] + used[m], rest, hipMemcpyDeviceToDevice, s_main));
          x.cur ^= 1;
        }
        x.have = rest;
      }
      if (final_piece) break;
      if (!progress) {
        err = "(ERROR) a record does not fit one piece of the stream, or the mates have different record counts";
        rc = SCALCE_ERR_FORMAT;
        goto fail_rc;
      }
    }
  }
  {
    double t0 = now_s();
    if ((rc = scalce_batch_order(b, s_main)) || (rc = scalce_batch_finish(b, s_main))) { err = scalce_last_error(ctx); goto fail_rc; }
    S.order_s = now_s() - t0;
    // the staging buffers are dead: give them back before the emit stage allocates the streams
    for (int m = 0; m < nm; m++)
      for (int i = 0; i < 2; i++) {
        hipFree(M[m].d_text[i]); M[m].d_text[i] = nullptr;
        hipFree(M[m].d_raw[i]); M[m].d_raw[i] = nullptr;
      }
    t0 = now_s();
    if ((rc = scalce_batch_emit(b, s_main)) || (rc = scalce_batch_finish(b, s_main))) { err = scalce_last_error(ctx); goto fail_rc; }
    S.emit_s = now_s() - t0;
    if (!(flags & SCALCE_STREAM_DEFER_ENTROPY)) {
      t0 = now_s();
      if ((rc = scalce_batch_entropy(b, nullptr, s_main)) || (rc = scalce_batch_finish(b, s_main))) { err = scalce_last_error(ctx); goto fail_rc; }
      S.entropy_s = now_s() - t0;
    }
  }
  S.total_s = now_s() - t_start;
  S.reads = scalce_batch_reads(b);
  goto cleanup;

fail:
  rc = SCALCE_ERR_HIP;
fail_rc:
  if (errbuf && errcap) snprintf(errbuf, errcap, "%s", err.c_str());
  if (b) scalce_batch_destroy(b);
  b = nullptr;
cleanup:
  for (int m = 0; m < nm; m++) {
    M[m].ring.shutdown();
    if (started && M[m].reader.joinable()) {
      // a reader blocked in rd() ends with its stream; one waiting for a free chunk has just been woken
      M[m].reader.join();
    }
    for (int i = 0; i < 2; i++) {
      if (M[m].d_text[i]) hipFree(M[m].d_text[i]);
      if (M[m].d_raw[i]) hipFree(M[m].d_raw[i]);
      if (M[m].up_ev[i]) hipEventDestroy(M[m].up_ev[i]);
      if (M[m].land_ev[i]) hipEventDestroy(M[m].land_ev[i]);
    }
    for (uint8_t *hp : M[m].pinned) hipHostFree(hp);
  }
  if (s_copy) hipStreamDestroy(s_copy);
  if (s_main) hipStreamDestroy(s_main);
  if (st) *st = S;
  *out = b;
  return rc;
}
