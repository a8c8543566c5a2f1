// pargz.hpp -- gzip input on several host threads (host code, no device work).
//
// The reference opens every input through zlib's gz reader (/root/reference/compress.cpp:756; FASTQ arrives gzipped), one
// thread, ~0.3 GB/s of text.  A gzip FILE is a sequence of members, and members are independent: bgzip (BGZF) files,
// concatenated .gz files and this repo's own -c gz containers (4 MiB members, cli.cpp) are made of many.  This reader
// takes the compressed file window by window (256 MiB), cuts a window into one segment per thread, and
//   1. every thread but the first looks for the first member that STARTS in its segment: a candidate is the gzip magic
//      with sane header bits (BGZF: the BC extra field says where the member ends, so the chain of headers is walked
//      instead); a candidate counts once the whole member inflates and its CRC-32 / length trailer checks out;
//   2. every thread inflates from its start to the next thread's start, members back to back;
//   3. the chain is checked -- each thread must END exactly where the next one started (a "member" found inside another
//      member's stored data would fail this) -- and where it does not, the rest of the window is redone by one thread.
// A file that is ONE member (plain `gzip`, pigz) has no inner starts: it streams through a single z_stream, as before.
// Behind a complete member, bytes that do not begin with the gzip magic end the stream, as they do for zlib's gzread
// (gz_look: "trailing garbage is ignored" -- zero padding of block-aligned or tape-written files); a member header that
// does begin with the magic but is damaged, or a damaged member body, is an error.  A member of more than MEMBER_CAP bytes
// of text is never held whole: the parallel window ends in front of it and the one stream takes it piece by piece.
// Output order is file order.  zlib only; nothing here knows about FASTQ.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace scalce_host {

class ParGz {
 public:
  ~ParGz() { close(); }
  bool open(const std::string &path, int threads) {
    close();
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) return false;
    struct stat st;
    if (fstat(fd_, &st) != 0) return false;
    size_ = (uint64_t)st.st_size;
    if (size_) {
      map_ = static_cast<const uint8_t *>(mmap(nullptr, size_, PROT_READ, MAP_SHARED, fd_, 0));
      if (map_ == MAP_FAILED) { map_ = nullptr; return false; }
      madvise(const_cast<uint8_t *>(map_), size_, MADV_SEQUENTIAL);
    }
    threads_ = std::max(1, std::min(threads, 64));
    pos_ = 0;
    memset(&zs_, 0, sizeof zs_);
    zs_open_ = false;
    members_done_ = false;
    parallel_windows = serial_bytes = 0;
    if (const char *e = getenv("SCALCE_PARGZ_MEMBER_CAP")) MEMBER_CAP = (size_t)std::max(1ll, atoll(e));
    return true;
  }
  void close() {
    if (zs_open_) { inflateEnd(&zs_); zs_open_ = false; }
    if (map_) { munmap(const_cast<uint8_t *>(map_), size_); map_ = nullptr; }
    if (fd_ >= 0) { ::close(fd_); fd_ = -1; }
    outs_.clear();
    cur_ = 0; cur_off_ = 0;
  }
  // up to cap bytes of text; 0 at the end of the file, -1 on a damaged stream
  int64_t read(void *dst, uint64_t cap) {
    for (;;) {
      while (cur_ < outs_.size() && cur_off_ == outs_[cur_].size()) { std::vector<uint8_t>().swap(outs_[cur_]); cur_++; cur_off_ = 0; }
      if (cur_ < outs_.size()) {
        const size_t k = (size_t)std::min<uint64_t>(cap, outs_[cur_].size() - cur_off_);
        memcpy(dst, outs_[cur_].data() + cur_off_, k);
        cur_off_ += k;
        return (int64_t)k;
      }
      outs_.clear(); cur_ = 0; cur_off_ = 0;
      if (!zs_open_ && pos_ >= size_) return 0;
      if (!fill()) return -1;
      if (outs_.empty() && !zs_open_ && pos_ >= size_) return 0;
    }
  }
  uint64_t parallel_windows = 0, serial_bytes = 0;  // diagnostics: windows inflated by several threads / text bytes through the one stream

 private:
  static constexpr uint64_t WINDOW = 256ull << 20;
  static constexpr size_t SERIAL_OUT = 64u << 20;
  size_t MEMBER_CAP = 1024u << 20;  // text bytes of one member a parallel window will hold (SCALCE_PARGZ_MEMBER_CAP: tests)
  bool members_done_ = false;  // at least one member has been inflated completely: a non-gzip tail is the end of the file
  int fd_ = -1;
  const uint8_t *map_ = nullptr;
  uint64_t size_ = 0, pos_ = 0;
  int threads_ = 1;
  z_stream zs_;
  bool zs_open_ = false;
  std::vector<std::vector<uint8_t>> outs_;
  size_t cur_ = 0, cur_off_ = 0;

  static bool has_magic(const uint8_t *p, uint64_t left) { return left >= 2 && p[0] == 0x1F && p[1] == 0x8B; }
  // the header of the member the stream stands at: what zlib itself insists on (method, reserved flag bits)
  static bool header_ok(const uint8_t *p, uint64_t left) {
    return left >= 18 && has_magic(p, left) && p[2] == 8 && !(p[3] & 0xE0);
  }
  // a CANDIDATE start found by scanning compressed bytes: also XFL as deflate writers set it (fewer false starts to inflate)
  static bool header_sane(const uint8_t *p, uint64_t left) {
    return header_ok(p, left) && (p[8] == 0 || p[8] == 2 || p[8] == 4);
  }
  // behind a complete member: anything that does not begin with the magic ends the file (zlib gz_look)
  bool at_tail_garbage() const { return members_done_ && pos_ < size_ && !has_magic(map_ + pos_, size_ - pos_); }
  // BGZF: extra subfield 'B','C' holds the member's size - 1
  static uint64_t bgzf_size(const uint8_t *p, uint64_t left) {
    if (!header_ok(p, left) || !(p[3] & 4)) return 0;
    const uint32_t xlen = p[10] | (p[11] << 8);
    if (12 + (uint64_t)xlen > left) return 0;
    for (uint32_t i = 0; i + 4 <= xlen;) {
      const uint8_t *f = p + 12 + i;
      const uint32_t len = f[2] | (f[3] << 8);
      if (f[0] == 'B' && f[1] == 'C' && len == 2) return (uint64_t)(f[4] | (f[5] << 8)) + 1;
      i += 4 + len;
    }
    return 0;
  }
  // one whole member at `at`, appended to out; returns the bytes it took, 0 if it is not a valid member
  // (give_up_at: stop once the member has produced more text than that; *gave_up says it was that and not damage)
  uint64_t inflate_member(uint64_t at, std::vector<uint8_t> &out, size_t give_up_at, bool *gave_up = nullptr) const {
    if (gave_up) *gave_up = false;
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) return 0;
    const size_t start = out.size();
    uint64_t in_done = 0;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
      const uint64_t in_left = size_ - at - in_done;
      if (!in_left) { rc = Z_DATA_ERROR; break; }
      z.next_in = const_cast<Bytef *>(map_ + at + in_done);
      z.avail_in = (uInt)std::min<uint64_t>(in_left, 1u << 30);
      const uInt fed = z.avail_in;
      if (out.size() == out.capacity() || out.capacity() - out.size() < (1u << 16)) out.reserve(std::max<size_t>(out.capacity() * 2, out.size() + (4u << 20)));
      const size_t room = out.capacity() - out.size();
      const size_t old = out.size();
      out.resize(out.capacity());
      z.next_out = out.data() + old;
      z.avail_out = (uInt)std::min<size_t>(room, 1u << 30);
      if (give_up_at) z.avail_out = (uInt)std::min<size_t>(z.avail_out, give_up_at + 1 - std::min(give_up_at, old - start));  // (never past the cap in one call)
      const uInt offered = z.avail_out;
      rc = inflate(&z, Z_NO_FLUSH);
      out.resize(old + (offered - z.avail_out));
      in_done += fed - z.avail_in;
      if (rc != Z_OK && rc != Z_STREAM_END) break;
      if (give_up_at && out.size() - start > give_up_at && rc != Z_STREAM_END) { rc = Z_DATA_ERROR; if (gave_up) *gave_up = true; break; }
    }
    inflateEnd(&z);
    if (rc != Z_STREAM_END) { out.resize(start); return 0; }
    return in_done;
  }
  // the first member that starts in [a, b): offset, or ~0; its text goes to `out`, *len = its compressed size
  uint64_t find_start(uint64_t a, uint64_t b, std::vector<uint8_t> &out, uint64_t *len) const {
    for (uint64_t c = a; c < b;) {
      const void *hit = memchr(map_ + c, 0x1F, (size_t)(b - c));
      if (!hit) break;
      c = (uint64_t)(static_cast<const uint8_t *>(hit) - map_);
      if (header_sane(map_ + c, size_ - c)) {
        const uint64_t k = inflate_member(c, out, std::min<size_t>(512u << 20, MEMBER_CAP));  // (a false start usually dies within a few bytes)
        if (k) { *len = k; return c; }
      }
      c++;
    }
    return ~0ull;
  }
  bool serial_step() {  // up to SERIAL_OUT bytes through the one stream; it stays open inside a member
    if (!zs_open_) {
      if (at_tail_garbage()) { pos_ = size_; return true; }
      memset(&zs_, 0, sizeof zs_);
      if (inflateInit2(&zs_, 15 + 16) != Z_OK) return false;
      zs_open_ = true;
    }
    std::vector<uint8_t> out(SERIAL_OUT);
    size_t got = 0;
    while (got < out.size()) {
      if (pos_ >= size_) return false;  // ends inside a member
      zs_.next_in = const_cast<Bytef *>(map_ + pos_);
      zs_.avail_in = (uInt)std::min<uint64_t>(size_ - pos_, 1u << 30);
      const uInt fed = zs_.avail_in;
      zs_.next_out = out.data() + got;
      zs_.avail_out = (uInt)(out.size() - got);
      const uInt offered = zs_.avail_out;
      const int rc = inflate(&zs_, Z_NO_FLUSH);
      got += offered - zs_.avail_out;
      pos_ += fed - zs_.avail_in;
      if (rc == Z_STREAM_END) { inflateEnd(&zs_); zs_open_ = false; members_done_ = true; break; }  // a member boundary: the next window may go parallel
      if (rc != Z_OK) return false;
    }
    out.resize(got);
    serial_bytes += got;
    outs_.push_back(std::move(out));
    return true;
  }
  bool fill() {
    if (zs_open_ || threads_ == 1) return serial_step();
    if (pos_ >= size_) return true;
    if (at_tail_garbage()) { pos_ = size_; return true; }
    if (!header_ok(map_ + pos_, size_ - pos_)) return serial_step();  // (zlib decides: a short or odd header is its to judge)
    const uint64_t wend = std::min(size_, pos_ + WINDOW);
    const int T = (int)std::min<uint64_t>((uint64_t)threads_, std::max<uint64_t>(1, (wend - pos_) >> 20));
    std::vector<uint64_t> start(T, ~0ull), first_len(T, 0), end(T, 0);
    std::vector<std::vector<uint8_t>> out(T);
    std::vector<char> ok(T, 1), stopped(T, 0);
    start[0] = pos_;
    const uint64_t seg = (wend - pos_ + T - 1) / T;
    const bool bgzf = bgzf_size(map_ + pos_, size_ - pos_) != 0;
    if (bgzf) {  // walk the chain of headers: a thread starts at the first member at or behind its segment's begin
      uint64_t at = pos_;
      int next = 1;
      while (at < wend && next < T) {
        const uint64_t k = bgzf_size(map_ + at, size_ - at);
        if (!k) break;
        at += k;
        while (next < T && pos_ + seg * (uint64_t)next <= at) { if (at < wend) start[next] = at; next++; }
      }
    } else if (T > 1) {
      std::vector<std::thread> pool;
      for (int i = 1; i < T; i++)
        pool.emplace_back([&, i]() {
          const uint64_t a = pos_ + seg * (uint64_t)i, b = std::min(wend, a + seg);
          if (a < b) start[i] = find_start(a, b, out[i], &first_len[i]);
        });
      for (auto &t : pool) t.join();
    }
    int live = 0;
    for (int i = 0; i < T; i++) live += start[i] != ~0ull;
    if (live < 2) {  // one member spans the window (or the file is small): the one stream
      return serial_step();
    }
    // from every start to the next one
    std::vector<int> idx;
    for (int i = 0; i < T; i++) if (start[i] != ~0ull) idx.push_back(i);
    std::vector<std::thread> pool;
    for (size_t q = 0; q < idx.size(); q++)
      pool.emplace_back([&, q]() {
        const int i = idx[q];
        const uint64_t stop = q + 1 < idx.size() ? start[idx[q + 1]] : wend;
        uint64_t at = start[i];
        if (first_len[i]) at += first_len[i];  // (its first member is already in out[i])
        while (at < stop && at < size_) {
          if (at != start[i] && !has_magic(map_ + at, size_ - at)) { stopped[i] = 1; break; }  // tail garbage behind a member: the next fill ends the file
          bool big = false;
          const uint64_t k = inflate_member(at, out[i], MEMBER_CAP, &big);
          if (!k) { if (big) stopped[i] = 1; else ok[i] = 0; break; }  // too large to hold whole: the one stream takes it from here
          at += k;
        }
        end[i] = at;
      });
    for (auto &t : pool) t.join();
    // the chain: thread q must end where thread q + 1 began
    size_t good = 0;
    const uint64_t pos0 = pos_;
    for (; good < idx.size(); good++) {
      const int i = idx[good];
      if (!ok[i]) {  // damage: what the thread inflated before it is good (and is kept), the one stream reports the rest
        if (end[i] > start[i]) { outs_.push_back(std::move(out[i])); pos_ = end[i]; members_done_ = true; good++; }
        break;
      }
      outs_.push_back(std::move(out[i]));
      pos_ = end[i];
      if (end[i] > start[i]) members_done_ = true;
      if (stopped[i] || (good + 1 < idx.size() && end[i] != start[idx[good + 1]])) { good++; break; }
    }
    if (pos_ == pos0) return serial_step();  // the member at pos_ itself is too large for a window, or damaged: the one stream
    parallel_windows++;
    return true;  // (whatever was not taken is redone from pos_ by the next fill)
  }
};

}  // namespace scalce_host
