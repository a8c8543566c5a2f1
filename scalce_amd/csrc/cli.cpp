// cli.cpp -- `scalce`: command-line drop-in for the reference's process contract
// (/root/reference/main.cpp:187-303, HELP) over libscalce_hip.so.
//
//   scalce [opts] -o PREFIX in_1.fastq[.gz] ...     ->  PREFIX_1.scalce{n,r,q} (+ PREFIX_2.* with -r)
//   scalce X_1.scalcen -d -o OUT                    ->  OUT_1.fastq (+ OUT_2.fastq)
//
// Host side only: option parsing, file I/O (plain / gzip via zlib), the quality sample
// (quality_mapping_init's read loop, qualities.cpp:64-97), the file headers of
// combine_and_compress_with_split (compress.cpp:289-343) and, for -d, the record loop of
// decompress.cpp:259-369.  Every byte of the hot path is produced by the HIP kernels; there is no CPU
// fallback -- without a GPU the tool stops with an error.
#include <getopt.h>
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <sys/time.h>
#include <zlib.h>
#include <atomic>
#include <memory>
#include <thread>
#include <algorithm>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <cerrno>
#include <csignal>

#include "../../include/scalce_hip.h"
#include "pargz.hpp"


#ifndef SCALCE_VERSION
#define SCALCE_VERSION "2.8-mi355x"
#endif

static void LOG(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
}
[[noreturn]] static void FAIL(const char *fmt, ...) {  // ERROR(), const.h:77-81
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "(ERROR) ");
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  exit(1);
}
static double now() {
  struct timeval t;
  gettimeofday(&t, 0);
  return t.tv_sec + 1e-6 * t.tv_usec;
}

struct Options {
  int lossy = 0, sample = 100000, threads = 0, split = 0;
  bool paired = false, use_names = true, no_ac = false, decompress = false;
  uint64_t bucket_set_size = 4ull << 30;  // main.cpp:68
  std::string out, library, patterns, temp = "__temp__", patterns_bin;
  int gpus = 1;                           // --gpus N: one process per GPU, ONE archive (plain-text input, -c no)
  int container = 1;                      // 0 plain, 1 gzip (main.cpp:181-184 at -T 1)
  uint64_t first_file_bytes[2] = {0, 0};  // --gpus over several files written out as one: where the first file ends (the
                                          // quality sample never leaves files[0], compress.cpp:761)
};

static const char *HELP_TEXT =
    "SCALCE " SCALCE_VERSION " (MI355X hot path)\n"
    "usage: scalce [options] -o OUTPUT FILE_1.fastq[.gz] ...      compress\n"
    "       scalce FILE_1.scalcen -d -o OUTPUT                   decompress\n"
    "  -o, --output STR            output prefix (required)\n"
    "  -r, --paired-end            FILE_1 is paired with the file whose last '1' is a '2'\n"
    "  -n, --skip-names STR        drop read names, regenerate them as STR.<index>\n"
    "  -c, --compression STR       container of the read/name streams: gz (default), pigz (= gz), no; bz is not built\n"
    "  -A, --no-arithmetic         store qualities raw instead of arithmetic coding\n"
    "  -p, --lossy-percentage INT  lossy quality transform, 0..100 (default 0)\n"
    "  -s, --sample-size INT       records sampled for the quality model (default 100000)\n"
    "  -B, --bucket-set-size NUM[M|G]  bucket storage that triggers a spill chunk (default 4G); order follows the reference\n"
    "  -P, --patterns FILE         text list of cores instead of the built-in table\n"
    "  -T, --threads INT           host threads that read plain input and deflate the gz containers (default: cores - 1,\n"
    "                              at most 64);\n"
    "                              the hot path itself runs on the GPU\n"
    "  -t, --temp-directory STR    accepted for compatibility (nothing is spilled)\n"
    "  -S, --split-reads INT       decompression: reads per output part\n"
    "  -d, --decompress    -v, --version    -h, --help\n"
    "      --gpus N                compress on N GPUs, one process each, into ONE archive that is byte for byte the archive of\n"
    "                              one GPU (and of the reference at -T 1 with the same -B); input: one plain FASTQ file (pair), -c no\n"
    "core table: --patterns-bin FILE or $SCALCE_PATTERNS or patterns.bin next to the executable\n";

// ---- small I/O helpers ------------------------------------------------------------------------------
static int g_threads_gz();
static std::vector<uint8_t> read_maybe_gz(const std::string &path) {  // IO_GZIP reader (compress.cpp:756); plain files pass through
  {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) FAIL("Cannot read file %s\n", path.c_str());
    uint8_t mg[2] = {0, 0};
    const bool gz = ::pread(fd, mg, 2, 0) == 2 && mg[0] == 0x1F && mg[1] == 0x8B;
    std::vector<uint8_t> out;
    if (!gz) {
      struct stat st;
      if (fstat(fd, &st) != 0) FAIL("Cannot read file %s\n", path.c_str());
      out.resize((size_t)st.st_size);
      size_t done = 0;
      while (done < out.size()) { const ssize_t k = ::pread(fd, out.data() + done, out.size() - done, (off_t)done); if (k <= 0) FAIL("Read error on %s\n", path.c_str()); done += (size_t)k; }
      ::close(fd);
      return out;
    }
    ::close(fd);
  }
  scalce_host::ParGz z;  // members inflated by several threads (our own -c gz containers are 4 MiB members)
  if (!z.open(path, g_threads_gz())) FAIL("Cannot read file %s\n", path.c_str());
  std::vector<uint8_t> out;
  std::vector<uint8_t> chunk(64u << 20);
  for (;;) {
    const int64_t k = z.read(chunk.data(), chunk.size());
    if (k < 0) FAIL("Read error on %s\n", path.c_str());
    if (k == 0) break;
    out.insert(out.end(), chunk.begin(), chunk.begin() + k);
  }
  return out;
}
static int g_threads_io();
// plain files with several threads (pread), gzip through zlib
static std::vector<uint8_t> read_file_fast(const std::string &path) {
  int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) FAIL("Cannot read file %s\n", path.c_str());
  uint8_t mg[2] = {0, 0};
  struct stat st;
  if (::pread(fd, mg, 2, 0) == 2 && mg[0] == 0x1F && mg[1] == 0x8B) { ::close(fd); return read_maybe_gz(path); }
  if (fstat(fd, &st) != 0) FAIL("Cannot read file %s\n", path.c_str());
  std::vector<uint8_t> out((size_t)st.st_size);
  const uint64_t n = out.size(), SL = 64u << 20;
  std::atomic<uint64_t> next{0};
  std::atomic<bool> bad{false};
  auto work = [&]() {
    for (uint64_t a; (a = next.fetch_add(SL)) < n;) {
      uint64_t done = a;
      const uint64_t b = std::min(n, a + SL);
      while (done < b) {
        const ssize_t k = ::pread(fd, out.data() + done, (size_t)(b - done), (off_t)done);
        if (k <= 0) { bad = true; return; }
        done += (uint64_t)k;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < g_threads_io(); t++) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
  ::close(fd);
  if (bad) FAIL("Read error on %s\n", path.c_str());
  return out;
}
static bool second_file(const std::string &p, std::string &out) {  // get_second_file, const.cpp:51-64
  out = p;
  for (int i = (int)out.size() - 1; i >= 0; i--)
    if (out[i] == '1') { out[i] = '2'; return true; }
  return false;
}
// Output file.  Plain: stdio.  gzip container (-c gz / pigz): the reference hands the stream to a pigz child or to
// zlib's gzwrite (buffio.cpp:148-188, 190-260); here the bytes are collected and deflated at close() by g_threads
// host threads, 4 MiB per independent gzip member -- concatenated members are one valid gzip file, which the
// reference's reader (gzread, decompress.cpp:99-113) and ours accept alike (once the hot path is on
// the GPU, single-threaded deflate of .scalcer/.scalcen is what the wall clock of a run is made of).
static int g_threads = 1;
static int g_threads_io() { return std::max(1, std::min(g_threads, 8)); }
static int g_threads_gz() { return std::max(1, std::min(g_threads, 32)); }
struct OutFile {
  bool gz = false;
  FILE *f = nullptr;
  std::vector<uint8_t> pending;  // gz only: bytes not deflated yet
  uint64_t written = 0;
  static constexpr size_t MEMBER = 4u << 20;
  void open(const std::string &path, bool gz_) {
    gz = gz_;
    f = path == "-" ? stdout : fopen(path.c_str(), "wb");
    if (!f) FAIL("Cannot create %s\n", path.c_str());
  }
  void raw(const uint8_t *b, size_t n) {
    written += n;
    while (n) {
      size_t k = n > (1u << 30) ? (1u << 30) : n;
      if (fwrite(b, 1, k, f) != k) FAIL("write failed\n");
      b += k; n -= k;
    }
  }
  void write(const void *p, size_t n) {
    const uint8_t *b = static_cast<const uint8_t *>(p);
    if (!gz) { raw(b, n); return; }
    pending.insert(pending.end(), b, b + n);
    // enough for every thread: deflate what is there, member by member (the stream never has to sit in memory whole)
    if (pending.size() >= MEMBER * (size_t)std::max(1, g_threads) * 2) flush_members(false);
  }
  // The reference writes its containers at zlib's default level (buffio.cpp: gzopen(path, "wb")), and so does this writer for
  // what deflate shrinks (names: to a third).  The read stream is 2-bit packed bases, as good as incompressible (the
  // reference's own gz gains 2.5 % on it) and the slowest thing zlib can be fed: ~20 MB/s per core at level 6 -- 3.6 of the
  // 5.8 s of a 50 M-read run with -c gz.  A member none of whose sample windows (eight of 16 KiB, spread over the member)
  // shrinks by a tenth at level 1 is therefore Huffman-coded only (Z_HUFFMAN_ONLY: no match search, ~15 x the speed, within
  // a percent of the size).  SCALCE_GZ_LEVEL=<n> turns that off and deflates every member at level n.
  static bool hardly_compressible(const uint8_t *src, size_t n) {
    static const bool off = getenv("SCALCE_GZ_LEVEL") != nullptr;   // (a level asked for by hand is applied to every member)
    if (off || n < 4096) return false;
    const size_t win = std::min<size_t>(n, 16u << 10), nwin = n >= 8 * win ? 8 : 1;
    std::vector<uint8_t> tmp(compressBound((uLong)win));
    for (size_t i = 0; i < nwin; i++) {
      const size_t at = nwin == 1 ? 0 : (n - win) / (nwin - 1) * i;
      uLongf got = (uLongf)tmp.size();
      if (compress2(tmp.data(), &got, src + at, (uLong)win, 1) != Z_OK) return false;
      if (got * 10 < win * 9) return false;   // this part does shrink: the member goes through deflate proper
    }
    return true;
  }
  static void deflate_member(const uint8_t *src, size_t n, std::vector<uint8_t> &out) {
    z_stream z;
    memset(&z, 0, sizeof z);
    const bool fast = n && hardly_compressible(src, n);
    // (what does shrink -- names -- goes at zlib's default level, the reference's: level 3 is 2.4 x the speed for a tenth more
    //  bytes of that stream; SCALCE_GZ_LEVEL=3 asks for it)
    static const int level = getenv("SCALCE_GZ_LEVEL") ? atoi(getenv("SCALCE_GZ_LEVEL")) : Z_DEFAULT_COMPRESSION;
    if (deflateInit2(&z, fast ? 1 : level, Z_DEFLATED, 15 + 16, 8, fast ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY) != Z_OK) FAIL("deflateInit2 failed\n");
    out.resize(deflateBound(&z, (uLong)n) + 64);
    z.next_in = const_cast<Bytef *>(src); z.avail_in = (uInt)n;
    z.next_out = out.data(); z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) FAIL("deflate failed\n");
    out.resize(z.total_out);
    deflateEnd(&z);
  }
  void flush_members(bool all) {
    const size_t whole = pending.size() / MEMBER, nchunks = all ? (pending.empty() ? 0 : (pending.size() + MEMBER - 1) / MEMBER) : whole;
    if (!nchunks) return;
    std::vector<std::vector<uint8_t>> members(nchunks);
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (size_t i; (i = next.fetch_add(1)) < nchunks;) {
        const size_t a = i * MEMBER, b = std::min(pending.size(), a + MEMBER);
        deflate_member(pending.data() + a, b - a, members[i]);
      }
    };
    const int nt = (int)std::min<size_t>((size_t)std::max(1, g_threads), nchunks);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    for (auto &m : members) raw(m.data(), m.size());
    const size_t done = std::min(pending.size(), nchunks * MEMBER);
    pending.erase(pending.begin(), pending.begin() + done);
  }
  void close() {
    if (gz && f) {
      const bool nothing = written == 0 && pending.empty();
      flush_members(true);
      if (nothing) { std::vector<uint8_t> m; deflate_member(nullptr, 0, m); raw(m.data(), m.size()); }  // an empty gzip file
      pending.clear(); pending.shrink_to_fit();
    }
    if (f && f != stdout) fclose(f);
    f = nullptr;
  }
};
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) FAIL("%s: %s\n", #x, hipGetErrorString(e_)); } while (0)
#define SCOK(ctx, x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s\n", scalce_last_error(ctx)); exit(1); } } while (0)

static std::vector<uint8_t> fetch(scalce_ctx *ctx, scalce_batch *b, int which, int mate) {
  const void *d = nullptr;
  uint64_t n = 0;
  SCOK(ctx, scalce_batch_output(b, which, mate, &d, &n));
  std::vector<uint8_t> v((size_t)n);
  if (n) SCOK(ctx, scalce_memcpy_d2h(ctx, v.data(), d, n));
  return v;
}

// A device stream to a file: slices come down into two pinned buffers in turn, the write (or deflate) of one slice runs
// while the next is on its way.  ONE writer per file: write(2) to tmpfs moves 6 GB/s from one thread on the bench host and
// holds the inode's lock -- four threads with a slice each and pwrite at the slices' offsets measured 0.9 s against 0.5 for
// the 1.9 GB of a quality stream, threads copying into a shared mapping 3-4 GB/s (tools/write_bench.cpp).  The framing
// kernel's stores into the pinned slice are not what limits the quality stream either: framed into HBM and brought down by
// the copy engine, the same 0.5 s.
struct Downloader {
  static constexpr size_t SLICE = 64u << 20;
  uint8_t *pin[2] = {nullptr, nullptr};
  hipStream_t s = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  Downloader() {
    HIPOK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) { HIPOK(hipHostMalloc(reinterpret_cast<void **>(&pin[i]), SLICE, hipHostMallocDefault)); HIPOK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
  }
  ~Downloader() {
    for (int i = 0; i < 2; i++) { if (pin[i]) hipHostFree(pin[i]); if (ev[i]) hipEventDestroy(ev[i]); }
    if (s) hipStreamDestroy(s);
  }
  void to_file(scalce_ctx *ctx, scalce_batch *b, int which, int mate, OutFile &f) {
    const void *d = nullptr;
    uint64_t n = 0;
    SCOK(ctx, scalce_batch_output(b, which, mate, &d, &n));
    range_to_file(static_cast<const uint8_t *>(d), n, f);
  }
  // the coded quality stream: framed on its way into the pinned slices (scalce_batch_qual_window), no copy in HBM
  void qual_to_file(scalce_ctx *ctx, scalce_batch *b, int mate, OutFile &f) {
    uint64_t total = 0;
    SCOK(ctx, scalce_batch_qual_bytes(b, mate, &total));
    const uint64_t nslices = (total + SLICE - 1) / SLICE;
    auto start = [&](uint64_t i) {
      const uint64_t off = i * SLICE, k = std::min<uint64_t>(SLICE, total - off);
      SCOK(ctx, scalce_batch_qual_window(b, mate, off, k, pin[i & 1], s));
      HIPOK(hipEventRecord(ev[i & 1], s));
    };
    if (nslices) start(0);
    for (uint64_t i = 0; i < nslices; i++) {
      HIPOK(hipEventSynchronize(ev[i & 1]));
      if (i + 1 < nslices) start(i + 1);
      f.write(pin[i & 1], (size_t)std::min<uint64_t>(SLICE, total - i * SLICE));
    }
  }
  void range_to_file(const uint8_t *src, uint64_t n, OutFile &f) {
    const uint64_t nslices = (n + SLICE - 1) / SLICE;
    auto start = [&](uint64_t i) {
      const uint64_t off = i * SLICE, k = std::min<uint64_t>(SLICE, n - off);
      HIPOK(hipMemcpyAsync(pin[i & 1], src + off, k, hipMemcpyDeviceToHost, s));
      HIPOK(hipEventRecord(ev[i & 1], s));
    };
    if (nslices) start(0);
    for (uint64_t i = 0; i < nslices; i++) {
      HIPOK(hipEventSynchronize(ev[i & 1]));
      if (i + 1 < nslices) start(i + 1);
      f.write(pin[i & 1], (size_t)std::min<uint64_t>(SLICE, n - i * SLICE));
    }
  }
};

// One mate's input: the files of the command line one after the other (compress.cpp:756-797 runs them through the same
// trie; the record stream is their concatenation), each plain or gzip (the reference opens everything through zlib,
// :767-779).  Plain files are read with read(2) straight into the pinned chunk the streaming host hands in.
struct MateSource {
  std::vector<std::string> files;
  size_t cur = 0;
  int fd = -1;
  scalce_host::ParGz pgz;     // gzip input: members inflated on several threads (pargz.hpp)
  bool gz = false;
  int gz_threads = 0;         // 0: g_threads (two mates read at once: the paired reader halves it)
  uint64_t gz_parallel_windows = 0, gz_serial_bytes = 0;
  std::vector<uint8_t> peek;  // bytes read ahead for the quality sample, served first
  size_t peek_pos = 0;
  bool all_plain = true;
  bool open_next() {
    while (cur < files.size()) {
      const std::string &path = files[cur++];
      fd = ::open(path.c_str(), O_RDONLY);
      if (fd < 0) FAIL("Cannot read file %s\n", path.c_str());
      uint8_t mg[2] = {0, 0};
      const ssize_t k = ::pread(fd, mg, 2, 0);
      if (k == 2 && mg[0] == 0x1F && mg[1] == 0x8B) {
        all_plain = false;
        ::close(fd);
        fd = -1;
        if (!pgz.open(path, gz_threads > 0 ? gz_threads : std::max(1, g_threads))) FAIL("Cannot read file %s\n", path.c_str());
        gz = true;
      } else {
#ifdef POSIX_FADV_SEQUENTIAL
        posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
      }
      return true;
    }
    return false;
  }
  uint64_t fpos = 0;  // plain files: where the next read starts
  bool hold_at_file_end = false;  // fill_peek: the sample never runs into the next file
  // The reference reads every file on its own, line by line (compress.cpp:756-811, gzgets): a file whose last line has no
  // newline still ends there.  Concatenated byte-wise that line would run into the next file's '@name', so a file that
  // does not end in a newline is given one.
  uint8_t last_byte = '\n';
  bool pending_newline = false;
  int64_t read_raw(void *dst, uint64_t cap) {
    for (;;) {
      if (pending_newline && cap) { pending_newline = false; last_byte = '\n'; *static_cast<uint8_t *>(dst) = '\n'; return 1; }
      if (fd < 0 && !gz) {
        if (hold_at_file_end && cur >= 1) return 0;
        if (!open_next()) return 0;
        fpos = 0;
      }
      int64_t k;
      if (gz) k = pgz.read(dst, cap);
      else k = read_plain(static_cast<uint8_t *>(dst), cap);
      if (k < 0) return -1;
      if (k > 0) { last_byte = static_cast<uint8_t *>(dst)[k - 1]; return k; }
      if (gz) { gz_parallel_windows += pgz.parallel_windows; gz_serial_bytes += pgz.serial_bytes; pgz.close(); gz = false; } else { ::close(fd); fd = -1; }
      if (last_byte != '\n') pending_newline = true;
    }
  }
  // A big request on a plain file is cut into slices read by several threads at once (pread): one thread copying out
  // of the page cache delivers about 5 GB/s, a tenth of what the upload behind it can take.
  int64_t read_plain(uint8_t *dst, uint64_t cap) {
    const uint64_t SLICE = 16u << 20;
    const int nt = (int)std::min<uint64_t>((uint64_t)std::max(1, std::min(g_threads, 8)), (cap + SLICE - 1) / SLICE);
    if (nt <= 1) {
      const int64_t k = ::pread(fd, dst, (size_t)std::min<uint64_t>(cap, 1u << 30), (off_t)fpos);
      if (k > 0) fpos += (uint64_t)k;
      return k;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) return -1;
    const uint64_t left = (uint64_t)st.st_size > fpos ? (uint64_t)st.st_size - fpos : 0;
    const uint64_t want = std::min(cap, left);
    if (!want) return 0;
    const uint64_t per = ((want + nt - 1) / nt + 4095) & ~4095ull;
    std::atomic<bool> bad{false};
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++) {
      const uint64_t a = (uint64_t)t * per, b = std::min(want, a + per);
      if (a >= b) break;
      pool.emplace_back([&, a, b]() {
        uint64_t done = a;
        while (done < b) {
          const ssize_t k = ::pread(fd, dst + done, (size_t)(b - done), (off_t)(fpos + done));
          if (k <= 0) { bad = true; return; }
          done += (uint64_t)k;
        }
      });
    }
    for (auto &t : pool) t.join();
    if (bad) return -1;
    fpos += want;
    return (int64_t)want;
  }
  int64_t read(void *dst, uint64_t cap) {
    if (peek_pos < peek.size()) {
      const size_t k = (size_t)std::min<uint64_t>(cap, peek.size() - peek_pos);
      memcpy(dst, peek.data() + peek_pos, k);
      peek_pos += k;
      if (peek_pos == peek.size()) { std::vector<uint8_t>().swap(peek); peek_pos = 0; }
      return (int64_t)k;
    }
    return read_raw(dst, cap);
  }
  // read ahead until the text holds `records` records or the FIRST file ends: quality_mapping_init's sample is taken
  // from files[0] alone (get_quality_stats, compress.cpp:761; the loop of qualities.cpp:66-78 stops at its end)
  void fill_peek(int records) {
    hold_at_file_end = true;
    fill_peek_held(records);
    hold_at_file_end = false;
  }
  void fill_peek_held(int records) {
    size_t lines = 0, scanned = 0;
    for (;;) {
      const uint8_t *p = peek.data();
      while (scanned < peek.size() && lines < 4 * (size_t)records) {
        const void *nl = memchr(p + scanned, '\n', peek.size() - scanned);
        if (!nl) { scanned = peek.size(); break; }
        scanned = (size_t)((const uint8_t *)nl - p) + 1;
        lines++;
      }
      if (lines >= 4 * (size_t)records) return;
      const size_t old = peek.size(), step = 16u << 20;
      peek.resize(old + step);
      int64_t got = 0;
      while ((size_t)got < step) {
        const int64_t k = read_raw(peek.data() + old + got, step - (size_t)got);
        if (k < 0) FAIL("Read error\n");
        if (k == 0) break;
        got += k;
      }
      peek.resize(old + (size_t)got);
      if (!got) return;
    }
  }
  static int64_t read_cb(void *user, void *dst, uint64_t cap) { return static_cast<MateSource *>(user)->read(dst, cap); }
};

// sampling loop of quality_mapping_init (qualities.cpp:64-97) on the text read ahead
static void sample_stats(const std::vector<uint8_t> &t, int sample, int32_t stat[128], int &read_length) {
  memset(stat, 0, 128 * sizeof(int32_t));
  size_t pos = 0;
  for (int i = 0; i < sample; i++) {
    size_t e = pos;
    bool ok = true;
    for (int k = 0; k < 3 && ok; k++) {
      const void *nl = e < t.size() ? memchr(t.data() + e, '\n', t.size() - e) : nullptr;
      if (!nl) ok = false; else e = (const uint8_t *)nl - t.data() + 1;
    }
    if (!ok) break;
    const void *nl = e < t.size() ? memchr(t.data() + e, '\n', t.size() - e) : nullptr;
    if (!nl) break;
    size_t qe = (const uint8_t *)nl - t.data();
    for (size_t j = e; j < qe; j++) stat[t[j] & 127]++;
    read_length = (int)(qe - e);
    pos = qe + 1;
  }
}

static std::vector<uint8_t> load_core_table(const Options &o, const char *argv0, bool &is_text) {
  is_text = false;
  if (!o.patterns.empty()) { is_text = true; return read_maybe_gz(o.patterns); }
  std::vector<std::string> cand;
  if (!o.patterns_bin.empty()) cand.push_back(o.patterns_bin);
  if (const char *e = getenv("SCALCE_PATTERNS")) cand.push_back(e);
  std::string self = argv0;
  size_t sl = self.rfind('/');
  cand.push_back((sl == std::string::npos ? std::string(".") : self.substr(0, sl)) + "/patterns.bin");
  cand.push_back("patterns.bin");
  for (auto &c : cand) {
    struct stat st;
    if (stat(c.c_str(), &st) == 0) return read_maybe_gz(c);
  }
  FAIL("No core table: give --patterns-bin FILE, -P LIST or set SCALCE_PATTERNS (the reference embeds patterns.bin at link time)\n");
}

// ---- compress -----------------------------------------------------------------------------------------
static int do_compress(const Options &o, const std::vector<std::string> &files, scalce_ctx *ctx) {
  const double t0 = now();
  const int nm = o.paired ? 2 : 1;
  uint64_t original = 0;
  int32_t qhist[128];
  scalce_params p;
  scalce_params_default(&p);
  p.paired = o.paired; p.use_names = o.use_names; p.no_ac = o.no_ac; p.bucket_set_size = o.bucket_set_size;
  LOG("Preprocessing FASTQ files ...\n");
  MateSource src[2];
  for (int m = 0; m < nm; m++) {
    for (size_t F = 0; F < files.size(); F++) {
      std::string path = files[F];
      if (m && !second_file(files[F], path))
        FAIL("Cannot get file name for paired end for file %s. File should contain character 1.\n", files[F].c_str());
      struct stat st;
      if (stat(path.c_str(), &st) == 0) original += (uint64_t)st.st_size;
      src[m].files.push_back(path);
    }
    // get_quality_stats looks at the first file only (compress.cpp:761)
    int rl = 0;
    src[m].fill_peek(o.sample);
    sample_stats(src[m].peek, o.sample, qhist, rl);
    scalce_qmap_init(&p.qmap[m], qhist, o.lossy);
    p.read_len[m] = rl;
    LOG("\tPaired end #%d, quality offset: %d\n\t               read length: %d\n", m + 1, p.qmap[m].offset, rl);
  }
  if (p.read_len[0] <= 0) FAIL("Cannot determine the read length of %s\n", files[0].c_str());
  // rows to expect: exact enough for plain text (a record is 2 L + 6 bytes plus its name), unknown behind gzip
  uint64_t hint = 0;
  if (src[0].all_plain && !src[0].gz) {
    uint64_t bytes = 0;
    for (auto &f : src[0].files) { struct stat st; if (stat(f.c_str(), &st) == 0) bytes += (uint64_t)st.st_size; }
    hint = bytes / (2 * (uint64_t)p.read_len[0] + 8) + 64;
    for (auto &f : src[0].files) { int fd = ::open(f.c_str(), O_RDONLY); uint8_t mg[2] = {0, 0}; if (fd >= 0) { if (::pread(fd, mg, 2, 0) == 2 && mg[0] == 0x1F && mg[1] == 0x8B) hint = 0; ::close(fd); } }
  }
  uint64_t piece = 256ull << 20;  // per chunk; three of them are pinned per mate
  if (const char *e = getenv("SCALCE_PIECE_BYTES")) piece = strtoull(e, nullptr, 10);
  {  // small inputs: no point in pinning gigabytes
    uint64_t bytes = 0;
    for (auto &f : src[0].files) { struct stat st; if (stat(f.c_str(), &st) == 0) bytes += (uint64_t)st.st_size; }
    if (hint && bytes + (1u << 20) < piece) piece = bytes + (1u << 20);
  }
  scalce_batch *b = nullptr;
  scalce_stream_stats ss;
  char emsg[512] = "";
  const double t1 = now();
  if (scalce_stream_compress(ctx, &p, MateSource::read_cb, &src[0], nm == 2 ? MateSource::read_cb : nullptr, nm == 2 ? &src[1] : nullptr, piece,
                             hint, SCALCE_STREAM_LEAN | SCALCE_STREAM_DEFER_ENTROPY, &b, &ss, emsg, sizeof emsg)) {
    fprintf(stderr, "%s\n", emsg[0] ? emsg : scalce_last_error(ctx));
    exit(1);
  }
  const double t2 = now();
  const uint64_t N = scalce_batch_reads(b);
  LOG("\tDone with file %s, %llu reads found\n", files[0].c_str(), (unsigned long long)N);
  uint32_t st4[6] = {0, 0, 0, 0, 0, 0};
  scalce_batch_stats(b, st4);
  // the arithmetic coder starts on its own stream; the read and name streams come down and are written beside it
  // (a run of up to 2048 blocks is one launch that takes as long as ONE block's serial chain, about 0.3 s)
  hipStream_t s_ent = nullptr;
  HIPOK(hipStreamCreateWithFlags(&s_ent, hipStreamNonBlocking));
  SCOK(ctx, scalce_batch_set_frame_on_demand(b, 1));  // the coded blocks are framed on their way into the pinned slices
  SCOK(ctx, scalce_batch_entropy_begin(b, nullptr, s_ent));

  // final writer: headers of combine_and_compress_with_split (compress.cpp:263-343)
  const uint8_t magic[8] = {'s', 'c', 'a', 'l', 'c', 'e', '2', '2'};
  const bool gz = o.container == 1;
  uint64_t new_size = 0;
  char fn[4096];
  // the files of a mate are written by a thread of its own (paired runs: two mates, two sets of files, two threads)
  auto per_mate = [&](auto &&body) {
    if (nm == 1) { body(0); return; }
    std::vector<std::thread> ts;
    for (int m = 0; m < nm; m++) ts.emplace_back([&body, m]() { HIPOK(hipSetDevice(0)); body(m); });
    for (auto &t : ts) t.join();
  };
  std::unique_ptr<Downloader> downs[2];  // (a mate's pinned slices serve both of its passes)
  per_mate([&](int m) {
    downs[m].reset(new Downloader);
    Downloader &down = *downs[m];
    char fn[4096];
    OutFile fR, fN;
    snprintf(fn, sizeof fn, "%s_%d.scalcer", o.out.c_str(), m + 1); fR.open(fn, gz);
    const int32_t noac = o.no_ac, len32 = p.read_len[m];
    fR.write(magic, 8); fR.write(&noac, 4); fR.write(&len32, 4);
    down.to_file(ctx, b, SCALCE_OUT_READS, m, fR);
    fR.close();
    snprintf(fn, sizeof fn, "%s_%d.scalcen", o.out.c_str(), m + 1); fN.open(fn, gz);
    const uint8_t un = o.use_names ? 1 : 0;
    fN.write(magic, 8); fN.write(&un, 1);
    if (o.use_names) down.to_file(ctx, b, SCALCE_OUT_NAMES, 0, fN);  // mate 2 repeats mate 1's names (:450-454)
    else { const int64_t z = 0; fN.write(&z, 8); fN.write(o.library.data(), o.library.size()); }
    fN.close();
  });
  const double t2b = now();
  SCOK(ctx, scalce_batch_finish(b, s_ent));  // the coder is through: sizes of the coded streams, device error word
  const double t2c = now();
  per_mate([&](int m) {
    Downloader &down = *downs[m];
    char fn[4096];
    OutFile fQ;
    snprintf(fn, sizeof fn, "%s_%d.scalceq", o.out.c_str(), m + 1); fQ.open(fn, o.no_ac ? gz : false);  // :249
    const int64_t phred = p.qmap[0].offset;  // mate 1's offset for both (compress.cpp:294,816-817)
    fQ.write(magic, 8); fQ.write(&phred, 8);
    if (!o.no_ac) {
      auto tb = fetch(ctx, b, SCALCE_OUT_TABLE, m);
      fQ.write(tb.data(), tb.size());
      const uint64_t total = N * (uint64_t)p.read_len[m];
      fQ.write(&total, 8);
    }
    down.qual_to_file(ctx, b, m, fQ);
    fQ.close();
    downs[m].reset();
  });
  const double t2d = now();
  for (int m = 0; m < nm; m++)
    for (const char *ext : {"r", "q", "n"}) {
      snprintf(fn, sizeof fn, "%s_%d.scalce%s", o.out.c_str(), m + 1, ext);
      struct stat st;
      if (stat(fn, &st) == 0) new_size += (uint64_t)st.st_size;
    }
  hipStreamDestroy(s_ent);
  const void *dc = nullptr;
  uint64_t nc = 0;
  SCOK(ctx, scalce_batch_output(b, SCALCE_OUT_BUCKET_COUNTS, 0, &dc, &nc));
  uint64_t unbucketed = 0;
  if (nc >= 8) SCOK(ctx, scalce_memcpy_d2h(ctx, &unbucketed, (const uint8_t *)dc + nc - 8, 8));
  scalce_batch_destroy(b);
  const double t3 = now();
  LOG("Statistics:\n\tTotal number of reads: %llu\n\tRead length: first end %d\n", (unsigned long long)N, p.read_len[0]);
  if (o.paired) LOG("\t             second end %d\n", p.read_len[1]);
  LOG("\tUnbucketed reads count: %llu, bucketed percentage %.2lf\n", (unsigned long long)unbucketed,
      N ? 100.0 * (double)(N - unbucketed) / (double)N : 0.0);
  LOG("\tLossy percentage: %d\n", o.lossy);
  LOG("\tSpill chunks: %u, pieces streamed: %llu\n", st4[3], (unsigned long long)ss.rounds);
  LOG("\tTime elapsed: %.2f s (sample %.2f; stream %.2f = waiting for the reader %.2f + for uploads %.2f + ingest/count/tokenize %.2f; "
      "order %.2f, emit %.2f; reads+names down and written beside the coder %.2f, waiting for the coder %.2f, qualities down and written %.2f, device buffers released %.2f)\n",
      t3 - t0, t1 - t0, ss.total_s - ss.order_s - ss.emit_s, ss.read_wait_s, ss.h2d_wait_s, ss.front_s, ss.order_s, ss.emit_s,
      t2b - t2, t2c - t2b, t2d - t2c, t3 - t2d);
  LOG("\tOriginal size: %.2lfM, new size: %.2lfM, compression factor: %.2lf\n", original / (1024.0 * 1024.0),
      new_size / (1024.0 * 1024.0), new_size ? (double)original / (double)new_size : 0.0);
  return 0;
}

// ---- compress on several GPUs ---------------------------------------------------------------------------------
// One process per GPU (forked before anything touches a GPU), rank r takes the r-th part of the input file, the ranks talk
// RCCL (scalce_sharded_compress) and every rank writes its pieces of the ONE archive with pwrite at the offsets the
// run-wide bucket counts give.  SCALCE_COMM=shm makes all ranks share GPU 0 over the shared-memory rehearsal transport.
static uint64_t count_newlines(int fd, uint64_t a, uint64_t b, int threads) {
  std::atomic<uint64_t> total{0};
  const uint64_t step = 64u << 20;
  std::atomic<uint64_t> next{a};
  auto work = [&]() {
    std::vector<char> buf(8u << 20);
    for (uint64_t at; (at = next.fetch_add(step)) < b;) {
      const uint64_t end = std::min(b, at + step);
      uint64_t n = 0;
      for (uint64_t p = at; p < end;) {
        const ssize_t k = ::pread(fd, buf.data(), (size_t)std::min<uint64_t>(buf.size(), end - p), (off_t)p);
        if (k <= 0) FAIL("Read error\n");
        const char *q = buf.data(), *e = q + k;
        while ((q = static_cast<const char *>(memchr(q, '\n', (size_t)(e - q))))) { n++; q++; }
        p += (uint64_t)k;
      }
      total += n;
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
  return total;
}
// byte offset behind the `skip`-th newline at or after `from`
static uint64_t skip_lines(int fd, uint64_t from, uint64_t size, uint64_t skip) {
  std::vector<char> buf(8u << 20);
  uint64_t p = from;
  while (skip && p < size) {
    const ssize_t k = ::pread(fd, buf.data(), (size_t)std::min<uint64_t>(buf.size(), size - p), (off_t)p);
    if (k <= 0) FAIL("Read error\n");
    const char *q = buf.data(), *e = q + k;
    while (skip && (q = static_cast<const char *>(memchr(q, '\n', (size_t)(e - q))))) { skip--; q++; }
    if (!skip) return p + (uint64_t)(q - buf.data());
    p += (uint64_t)k;
  }
  return p;
}
static void pwrite_all(int fd, const void *src, size_t n, uint64_t off) {
  const uint8_t *p = static_cast<const uint8_t *>(src);
  while (n) {
    const ssize_t k = ::pwrite(fd, p, n, (off_t)off);
    if (k <= 0) FAIL("write failed\n");
    p += k; n -= (size_t)k; off += (uint64_t)k;
  }
}

static int rank_main(const Options &o, const std::vector<std::string> &files, const char *argv0, int rank, int world, const std::string &tag) {
  const bool shm = getenv("SCALCE_COMM") && !strcmp(getenv("SCALCE_COMM"), "shm");
  const int device = shm ? 0 : rank;
  const int nm = o.paired ? 2 : 1;
  const double t0 = now();
  scalce_ctx *ctx = nullptr;
  if (scalce_ctx_create(device, &ctx)) FAIL("%s\n", scalce_last_error(ctx));
  bool is_text = false;
  std::vector<uint8_t> table = load_core_table(o, argv0, is_text);
  if (is_text) SCOK(ctx, scalce_patterns_load_text(ctx, (const char *)table.data(), table.size()));
  else SCOK(ctx, scalce_patterns_load_bin(ctx, table.data(), table.size()));
  // communicator
  scalce_comm *comm = nullptr;
  if (shm) {
    if (scalce_comm_create_shm(device, world, rank, ("/scalce_cli_" + tag).c_str(), 1ull << 30, &comm)) FAIL("%s\n", scalce_comm_error(comm));
  } else {
    const std::string idfile = o.out + ".rcclid." + tag;
    uint8_t id[SCALCE_COMM_ID_BYTES];
    if (rank == 0) {
      if (scalce_comm_unique_id(id)) FAIL("RCCL is not available\n");
      FILE *f = fopen((idfile + ".tmp").c_str(), "wb");
      if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) FAIL("Cannot write %s\n", idfile.c_str());
      fclose(f);
      rename((idfile + ".tmp").c_str(), idfile.c_str());
    } else {
      FILE *f = nullptr;
      for (int tries = 0; tries < 60000 && !(f = fopen(idfile.c_str(), "rb")); tries++) usleep(1000);
      if (!f || fread(id, 1, sizeof id, f) != sizeof id) FAIL("rank 0 did not publish the RCCL id\n");
      fclose(f);
    }
    if (scalce_comm_create_rccl(device, world, rank, id, &comm)) FAIL("%s\n", scalce_comm_error(comm));
    scalce_comm_barrier(comm, nullptr);
    if (rank == 0) unlink(idfile.c_str());
  }
  // quality model: the first records of the FILE, the same on every rank (get_quality_stats, compress.cpp:761)
  scalce_params p;
  scalce_params_default(&p);
  p.paired = o.paired; p.use_names = o.use_names; p.no_ac = o.no_ac; p.bucket_set_size = o.bucket_set_size;
  std::string path[2] = {files[0], files[0]};
  if (o.paired && !second_file(files[0], path[1])) FAIL("Cannot get file name for paired end for file %s. File should contain character 1.\n", files[0].c_str());
  int fd[2] = {-1, -1};
  uint64_t fsize[2] = {0, 0};
  for (int m = 0; m < nm; m++) {
    MateSource src;
    src.files.push_back(path[m]);
    src.fill_peek(o.sample);
    if (!src.all_plain) FAIL("--gpus needs plain (not gzip) input: the file is split by byte ranges\n");
    if (o.first_file_bytes[m] && src.peek.size() > o.first_file_bytes[m]) src.peek.resize((size_t)o.first_file_bytes[m]);
    int32_t qhist[128];
    int rl = 0;
    sample_stats(src.peek, o.sample, qhist, rl);
    scalce_qmap_init(&p.qmap[m], qhist, o.lossy);
    p.read_len[m] = rl;
    fd[m] = ::open(path[m].c_str(), O_RDONLY);
    struct stat st;
    if (fd[m] < 0 || fstat(fd[m], &st) != 0) FAIL("Cannot read file %s\n", path[m].c_str());
    fsize[m] = (uint64_t)st.st_size;
    if (rank == 0) LOG("\tPaired end #%d, quality offset: %d\n\t               read length: %d\n", m + 1, p.qmap[m].offset, rl);
  }
  if (p.read_len[0] <= 0) FAIL("Cannot determine the read length of %s\n", files[0].c_str());
  // ---- this rank's records: line-aligned byte ranges, line counts of everybody, then cuts at multiples of four lines
  hipStream_t s = nullptr;
  HIPOK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint64_t *d_io = nullptr;
  HIPOK(hipMalloc(reinterpret_cast<void **>(&d_io), sizeof(uint64_t) * (64 + 64 * 4)));
  uint64_t lo[2], hi[2];
  std::vector<uint64_t> K;  // rank r starts at record K[r], in every mate
  for (int m = 0; m < nm; m++) {
    auto line_start = [&](int r) -> uint64_t {  // first line start at or behind the r-th N-th of the file
      if (r <= 0) return 0;
      if (r >= world) return fsize[m];
      const uint64_t at = fsize[m] / (uint64_t)world * (uint64_t)r;
      return at ? skip_lines(fd[m], at - 1, fsize[m], 1) : 0;
    };
    const uint64_t a = line_start(rank), b = line_start(rank + 1);
    uint64_t mine[2] = {count_newlines(fd[m], a, b, std::max(1, std::min(g_threads, 16))), a};
    std::vector<uint64_t> all((size_t)world * 2);
    HIPOK(hipMemcpy(d_io, mine, sizeof mine, hipMemcpyHostToDevice));
    if (scalce_comm_all_gather(comm, d_io, d_io + 64, sizeof mine, s)) FAIL("%s\n", scalce_comm_error(comm));
    HIPOK(hipStreamSynchronize(s));
    HIPOK(hipMemcpy(all.data(), d_io + 64, all.size() * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> first_line(world + 1, 0);  // run-wide line number at which every tentative range begins
    for (int r = 0; r < world; r++) first_line[r + 1] = first_line[r] + all[2 * r];
    if (first_line[world] % 4) FAIL("%s has %llu lines: not a multiple of 4\n", path[m].c_str(), (unsigned long long)first_line[world]);
    if (m == 1 && first_line[world] != 4 * K[world]) FAIL("mates have different record counts\n");
    // rank r starts at record K_r = ceil(first line of mate 1's r-th range / 4): the same record in every mate
    if (m == 0) { K.assign(world + 1, 0); for (int r = 0; r <= world; r++) K[r] = (first_line[r] + 3) / 4; }
    auto offset_of_line = [&](uint64_t line) -> uint64_t {
      if (line >= first_line[world]) return fsize[m];
      int r = 0;
      while (r + 1 < world && first_line[r + 1] <= line) r++;
      return skip_lines(fd[m], all[2 * r + 1], fsize[m], line - first_line[r]);
    };
    lo[m] = offset_of_line(4 * K[rank]);
    hi[m] = offset_of_line(4 * K[rank + 1]);
  }
  // ---- the piece into HBM
  uint8_t *d_text[2] = {nullptr, nullptr};
  {
    const size_t CH = 256u << 20;
    uint8_t *pin[2];
    hipEvent_t ev[2];
    for (int i = 0; i < 2; i++) { HIPOK(hipHostMalloc(reinterpret_cast<void **>(&pin[i]), CH, hipHostMallocDefault)); HIPOK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
    for (int m = 0; m < nm; m++) {
      const uint64_t n = hi[m] - lo[m];
      HIPOK(hipMalloc(reinterpret_cast<void **>(&d_text[m]), n + 256));
      MateSource src;  // (parallel pread into the pinned chunk)
      src.fd = fd[m];
      src.fpos = lo[m];
      uint64_t done = 0;
      for (int i = 0; done < n; i ^= 1) {
        HIPOK(hipEventSynchronize(ev[i]));
        const uint64_t k = std::min<uint64_t>(CH, n - done);
        uint64_t got = 0;
        while (got < k) { const int64_t r = src.read_plain(pin[i] + got, k - got); if (r <= 0) FAIL("Read error\n"); got += (uint64_t)r; }
        HIPOK(hipMemcpyAsync(d_text[m] + done, pin[i], k, hipMemcpyHostToDevice, s));
        HIPOK(hipEventRecord(ev[i], s));
        done += k;
      }
      src.fd = -1;
    }
    HIPOK(hipStreamSynchronize(s));
    for (int i = 0; i < 2; i++) { hipHostFree(pin[i]); hipEventDestroy(ev[i]); }
  }
  const double t1 = now();
  // ---- the sharded run
  scalce_batch *b = nullptr;
  const uint64_t rows = (hi[0] - lo[0]) / (2 * (uint64_t)p.read_len[0] + 7) + 64;
  SCOK(ctx, scalce_batch_create(ctx, &p, rows + rows / 3, std::max(hi[0] - lo[0], nm == 2 ? hi[1] - lo[1] : 0) + 256, &b));
  scalce_shard_result res;
  memset(&res, 0, sizeof res);
  {
    const int rc = scalce_sharded_compress(comm, ctx, b, d_text[0], hi[0] - lo[0], nm == 2 ? d_text[1] : nullptr, nm == 2 ? hi[1] - lo[1] : 0, 0, s, nullptr, &res);
    if (rc) exit(1);
  }
  for (int m = 0; m < nm; m++) hipFree(d_text[m]);
  const double t2 = now();
  // ---- every rank writes its pieces of the archive
  const uint32_t nb1 = res.nb1;
  std::vector<int32_t> bucket_pattern(nb1);
  int32_t nst = 0, nbk = 0;
  scalce_patterns_describe_host(table.data(), table.size(), is_text ? 1 : 0, bucket_pattern.data(), nb1, &nst, &nbk);
  std::vector<uint64_t> recsz(nb1), Cg(nb1, 0);
  const int L0 = p.read_len[0], sz_meta = L0 > 255 ? 2 : 1;
  for (uint32_t k = 0; k < nb1; k++) {
    const int lv = bucket_pattern[k] == SCALCE_ROOT_CORE ? 0 : scalce_pattern_length(ctx, bucket_pattern[k]);
    recsz[k] = (uint64_t)((L0 - lv + 3) / 4 + sz_meta);
    for (int r = 0; r < world; r++) Cg[k] += res.counts[(size_t)r * nb1 + k];
  }
  const uint8_t magic[8] = {'s', 'c', 'a', 'l', 'c', 'e', '2', '2'};
  const uint64_t N = res.reads_total;
  char fn[4096];
  auto open_out = [&](const char *ext, int m) {
    snprintf(fn, sizeof fn, "%s_%d.scalce%s", o.out.c_str(), m + 1, ext);
    if (rank == 0) { const int f = ::open(fn, O_CREAT | O_TRUNC | O_WRONLY, 0644); if (f < 0) FAIL("Cannot create %s\n", fn); ::close(f); }
    scalce_comm_barrier(comm, s);
    const int f = ::open(fn, O_WRONLY);
    if (f < 0) FAIL("Cannot open %s\n", fn);
    return f;
  };
  auto host_copy = [&](int which, int m) { return fetch(ctx, b, which, m); };
  for (int m = 0; m < nm; m++) {
    {  // .scalcer
      const int f = open_out("r", m);
      if (rank == 0) { const int32_t noac = o.no_ac, len32 = p.read_len[m]; uint8_t h[16]; memcpy(h, magic, 8); memcpy(h + 8, &noac, 4); memcpy(h + 12, &len32, 4); pwrite_all(f, h, 16, 0); }
      std::vector<uint8_t> mine = host_copy(SCALCE_OUT_READS, m);
      if (m == 0) {
        uint64_t base = 16, local = 0;
        for (uint32_t k = 0; k < nb1; k++) {
          if (!Cg[k]) continue;
          if (rank == 0) { uint8_t h[12]; const int32_t core = bucket_pattern[k]; memcpy(h, &core, 4); memcpy(h + 4, &Cg[k], 8); pwrite_all(f, h, 12, base); }
          uint64_t before = 0;
          for (int r = 0; r < rank; r++) before += res.counts[(size_t)r * nb1 + k];
          const uint64_t c = res.counts[(size_t)rank * nb1 + k];
          if (c) { pwrite_all(f, mine.data() + local + 12, c * recsz[k], base + 12 + before * recsz[k]); local += 12 + c * recsz[k]; }
          base += 12 + Cg[k] * recsz[k];
        }
      } else {  // mate 2: bare rows in mate 1's order
        const uint64_t w = (uint64_t)(p.read_len[1] + 3) / 4;
        uint64_t first = 0, local = 0;
        for (uint32_t k = 0; k < nb1; k++) {
          uint64_t before = 0;
          for (int r = 0; r < rank; r++) before += res.counts[(size_t)r * nb1 + k];
          const uint64_t c = res.counts[(size_t)rank * nb1 + k];
          if (c) { pwrite_all(f, mine.data() + local * w, c * w, 16 + (first + before) * w); local += c; }
          first += Cg[k];
        }
      }
      ::close(f);
    }
    {  // .scalcen (mate 2 repeats mate 1's names, compress.cpp:450-454)
      const int f = open_out("n", m);
      const uint8_t un = o.use_names ? 1 : 0;
      if (rank == 0) { uint8_t h[9]; memcpy(h, magic, 8); h[8] = un; pwrite_all(f, h, 9, 0); }
      if (o.use_names) {
        std::vector<uint8_t> mine = host_copy(SCALCE_OUT_NAMES, 0);
        uint64_t base = 9, local = 0;
        for (uint32_t k = 0; k < nb1; k++) {
          uint64_t before = 0, all = 0;
          for (int r = 0; r < world; r++) { if (r < rank) before += res.name_bytes[(size_t)r * nb1 + k]; all += res.name_bytes[(size_t)r * nb1 + k]; }
          const uint64_t c = res.name_bytes[(size_t)rank * nb1 + k];
          if (c) { pwrite_all(f, mine.data() + local, c, base + before); local += c; }
          base += all;
        }
      } else if (rank == 0) {
        const int64_t z = 0;
        pwrite_all(f, &z, 8, 9);
        pwrite_all(f, o.library.data(), o.library.size(), 17);
      }
      ::close(f);
    }
    {  // .scalceq
      const int f = open_out("q", m);
      uint64_t base = 16;
      if (rank == 0) { const int64_t phred = p.qmap[0].offset; uint8_t h[16]; memcpy(h, magic, 8); memcpy(h + 8, &phred, 8); pwrite_all(f, h, 16, 0); }
      if (!o.no_ac) {
        if (rank == 0) {
          std::vector<uint8_t> tb = host_copy(SCALCE_OUT_TABLE, m);
          pwrite_all(f, tb.data(), tb.size(), 16);
          const uint64_t total = N * (uint64_t)p.read_len[m];
          pwrite_all(f, &total, 8, 16 + tb.size());
        }
        base = 16 + 2048000 + 8;
        uint64_t before = 0;
        for (int r = 0; r < rank; r++) before += res.coded_bytes[m][r];
        std::vector<uint8_t> mine = host_copy(SCALCE_OUT_QUAL, m);
        pwrite_all(f, mine.data(), mine.size(), base + before);
      } else {  // -A: the raw q' rows, bucket by bucket in rank order (compress.cpp:389-390)
        std::vector<uint8_t> mine = host_copy(SCALCE_OUT_QSTREAM, m);
        const uint64_t w = (uint64_t)p.read_len[m];
        uint64_t first = 0, local = 0;
        for (uint32_t k = 0; k < nb1; k++) {
          uint64_t before = 0;
          for (int r = 0; r < rank; r++) before += res.counts[(size_t)r * nb1 + k];
          const uint64_t c = res.counts[(size_t)rank * nb1 + k];
          if (c) { pwrite_all(f, mine.data() + local * w, c * w, 16 + (first + before) * w); local += c; }
          first += Cg[k];
        }
      }
      ::close(f);
    }
  }
  scalce_comm_barrier(comm, s);
  const double t3 = now();
  if (rank == 0) {
    LOG("\tDone with file %s, %llu reads found\n", files[0].c_str(), (unsigned long long)N);
    LOG("Statistics:\n\tTotal number of reads: %llu\n\tRead length: first end %d\n", (unsigned long long)N, p.read_len[0]);
    if (o.paired) LOG("\t             second end %d\n", p.read_len[1]);
    uint64_t unb = Cg[nb1 - 1];
    LOG("\tUnbucketed reads count: %llu, bucketed percentage %.2lf\n", (unsigned long long)unb, N ? 100.0 * (double)(N - unb) / (double)N : 0.0);
    LOG("\tLossy percentage: %d\n", o.lossy);
    LOG("\tGPUs: %d, spill chunks: %u, tie-break rounds: %u\n", world, res.chunks_total, res.rounds);
    LOG("\tTime elapsed: %.2f s (split + read + upload %.2f, sharded hot path %.2f, download + write %.2f)\n", t3 - t0, t1 - t0, t2 - t1, t3 - t2);
  }
  scalce_shard_result_free(&res);
  scalce_batch_destroy(b);
  scalce_comm_destroy(comm);
  scalce_ctx_destroy(ctx);
  return 0;
}

// --gpus splits ONE plain file per mate by byte ranges.  Several input files, or gzip input (the reference runs any number
// of files, each through its gz reader, compress.cpp:756-811): the record stream of every mate -- the files one after the
// other, inflated -- is written out once as a plain file under -t, and the ranks split that.
static bool plain_single_input(const Options &o, const std::vector<std::string> &files) {
  if (files.size() != 1) return false;
  for (int m = 0; m < (o.paired ? 2 : 1); m++) {
    std::string path = files[0];
    if (m && !second_file(files[0], path)) return true;  // (the rank reports it)
    const int fd = ::open(path.c_str(), O_RDONLY);
    uint8_t mg[2] = {0, 0};
    const bool gz = fd >= 0 && ::pread(fd, mg, 2, 0) == 2 && mg[0] == 0x1F && mg[1] == 0x8B;
    if (fd >= 0) ::close(fd);
    if (gz) return false;
  }
  return true;
}
// files the parent of a --gpus run must not leave behind, whichever way it ends (FAIL() is exit(1): atexit runs; the ranks
// leave through _exit and never get here)
static std::vector<std::string> g_unlink_at_exit;
static void unlink_at_exit() { for (auto &f : g_unlink_at_exit) unlink(f.c_str()); }
static std::string materialize_inputs(Options &o, const std::vector<std::string> &files, const char *tag, std::vector<std::string> &made) {
  struct stat st;
  std::string dir = o.temp;
  // (no silent fall-back to /tmp: at full size this is 100+ GB, and the user said where temporary files go)
  if (stat(dir.c_str(), &st) != 0 && mkdir(dir.c_str(), 0777) != 0) FAIL("Cannot create temporary directory %s (-t)\n", dir.c_str());
  static bool registered = false;
  if (!registered) { atexit(unlink_at_exit); registered = true; }
  // the mate digit is the LAST '1' of the path (get_second_file, const.cpp:51-64): it closes the name
  std::string base = dir + "/scalce_gpus_" + tag + "_m";
  for (char &c : base) if (&c >= &base[dir.size()] && c == '1') c = 'x';
  std::vector<uint8_t> buf(64u << 20);
  for (int m = 0; m < (o.paired ? 2 : 1); m++) {
    MateSource src;
    for (auto &f : files) {
      std::string path = f;
      if (m && !second_file(f, path)) FAIL("Cannot get file name for paired end for file %s. File should contain character 1.\n", f.c_str());
      if (stat(path.c_str(), &st) != 0) FAIL("File %s does not exist or it is not accessible.\n", path.c_str());
      src.files.push_back(path);
    }
    const std::string out = base + (m ? "2" : "1");
    FILE *f = fopen(out.c_str(), "wb");
    if (!f) FAIL("Cannot create %s\n", out.c_str());
    made.push_back(out);
    g_unlink_at_exit.push_back(out);
    uint64_t total = 0;
    for (;;) {
      const size_t before = src.cur;
      const int64_t k = src.read(buf.data(), buf.size());
      if (k < 0) FAIL("Read error\n");
      // (a read never spans two files: the first file has ended when the source has moved on to the second)
      if (!o.first_file_bytes[m] && files.size() > 1 && before >= 1 && src.cur >= 2) o.first_file_bytes[m] = total;
      if (k == 0) break;
      if (fwrite(buf.data(), 1, (size_t)k, f) != (size_t)k) FAIL("write failed (%s)\n", out.c_str());
      total += (uint64_t)k;
    }
    fclose(f);
  }
  return base + "1";
}
// -c gz behind --gpus: the ranks write the plain archive at computed offsets; the streams the reference would have sent
// through its gzip writer are then rewritten as gzip members by this process's threads
static void gzip_in_place(const std::string &path) {
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) FAIL("Cannot read %s\n", path.c_str());
  OutFile out;
  out.open(path + ".gz.tmp", true);
  std::vector<uint8_t> buf(256u << 20);
  for (;;) {
    const ssize_t k = ::read(fd, buf.data(), buf.size());
    if (k < 0) FAIL("Read error on %s\n", path.c_str());
    if (k == 0) break;
    out.write(buf.data(), (size_t)k);
  }
  ::close(fd);
  out.close();
  if (rename((path + ".gz.tmp").c_str(), path.c_str()) != 0) FAIL("Cannot replace %s\n", path.c_str());
}

static int multi_gpu_compress(const Options &o_in, const std::vector<std::string> &files_in, const char *argv0) {
  if (o_in.gpus > 64) FAIL("--gpus: at most 64\n");
  LOG("Preprocessing FASTQ files ...\n");
  char tag[64];
  snprintf(tag, sizeof tag, "%d_%ld", (int)getpid(), (long)time(nullptr));
  {
    const int hw = (int)std::thread::hardware_concurrency();
    g_threads = o_in.threads > 0 ? o_in.threads : std::max(1, std::min(64, hw - 1));
  }
  Options o = o_in;
  o.container = 0;  // the ranks write plain; -c gz is applied to the finished files below
  std::vector<std::string> files = files_in, made;
  if (!plain_single_input(o, files)) {
    const double t0 = now();
    files.assign(1, materialize_inputs(o, files_in, tag, made));
    LOG("\t%zu input file(s) per mate written out as one plain file under %s (%.2f s)\n", files_in.size(),
        made[0].substr(0, made[0].rfind('/')).c_str(), now() - t0);
  }
  struct Cleanup {
    std::vector<std::string> &v;
    ~Cleanup() { for (auto &f : v) unlink(f.c_str()); }
  } cleanup{made};
  {
    const int hw = (int)std::thread::hardware_concurrency();
    g_threads = o.threads > 0 ? o.threads : std::max(1, std::min(64, hw / std::max(1, o.gpus)));
  }
  std::vector<pid_t> kids;
  for (int r = 0; r < o.gpus; r++) {
    const pid_t pid = fork();
    if (pid < 0) FAIL("fork failed\n");
    if (pid == 0) _exit(rank_main(o, files, argv0, r, o.gpus, tag));
    kids.push_back(pid);
  }
  // Ranks are collected in the order they end.  One that fails on its own (a crash, an error outside the collective
  // protocol of scalce_sharded_compress) leaves the others waiting for it in a collective: the first bad exit ends them.
  int bad = 0;
  size_t left = kids.size();
  while (left) {
    int st = 0;
    const pid_t k = waitpid(-1, &st, 0);
    if (k < 0) { if (errno == EINTR) continue; break; }
    auto it = std::find(kids.begin(), kids.end(), k);
    if (it == kids.end()) continue;
    *it = -1;
    left--;
    const bool ok = WIFEXITED(st) && WEXITSTATUS(st) == 0;
    if (!ok && !bad) {
      bad = 1;
      for (pid_t o2 : kids) if (o2 > 0) kill(o2, SIGTERM);
    }
  }
  if (bad) {  // what the ranks had written so far is not an archive: nothing stays behind under the output name
    for (int m = 0; m < (o.paired ? 2 : 1); m++)
      for (const char *ext : {"r", "n", "q"}) unlink((o.out + "_" + std::to_string(m + 1) + ".scalce" + ext).c_str());
    fprintf(stderr, "(ERROR) a rank failed\n");
    return 1;
  }
  if (o_in.container != 0) {  // compress.cpp:249: the arithmetic-coded stream is never containerised
    const int hw = (int)std::thread::hardware_concurrency();
    g_threads = o_in.threads > 0 ? o_in.threads : std::max(1, std::min(64, hw - 1));
    const double t0 = now();
    for (int m = 0; m < (o.paired ? 2 : 1); m++)
      for (const char *ext : {"r", "n", "q"}) {
        if (ext[0] == 'q' && !o.no_ac) continue;
        gzip_in_place(o.out + "_" + std::to_string(m + 1) + ".scalce" + ext);
      }
    LOG("\tgzip containers written by %d host threads: %.2f s\n", g_threads, now() - t0);
  }
  LOG("Done!\n");
  return 0;
}

// ---- decompress ------------------------------------------------------------------------------------------
static std::string scalce_name(std::string path, char c) {  // get_file_name, decompress.cpp:72-77
  size_t p = path.rfind(".scalce");
  if (p != std::string::npos && p + 7 < path.size()) path[p + 7] = c;
  return path;
}
struct Reader {
  std::vector<uint8_t> v;
  size_t pos = 0;
  size_t read(void *dst, size_t n) {
    size_t k = pos + n <= v.size() ? n : (v.size() - pos);
    memcpy(dst, v.data() + pos, k);
    pos += k;
    return k;
  }
};

static int do_decompress(const Options &o, const std::string &path, scalce_ctx *ctx) {
  const double t0 = now();
  const int nm = o.paired ? 2 : 1;
  std::string base[2] = {path, path};
  if (o.paired && !second_file(path, base[1])) FAIL("Cannot get file name for paired end for file %s.\n", path.c_str());
  Reader R[2], Q[2], Nn[2];
  int32_t len[2] = {0, 0}, no_ac = 0;
  int64_t phred[2] = {0, 0};
  {  // the archive's files side by side (each by its own pread threads): 54 GB of a 200 M-pair archive one after the other were
     // 10 of the run's 35 s
    std::vector<std::thread> rd;
    for (int m = 0; m < nm; m++) {
      rd.emplace_back([&, m]() { R[m].v = read_file_fast(scalce_name(base[m], 'r')); });  // container sniffing (decompress.cpp:99-113): gzip magic or plain
      rd.emplace_back([&, m]() { Nn[m].v = read_file_fast(scalce_name(base[m], 'n')); });
      rd.emplace_back([&, m]() { Q[m].v = read_file_fast(scalce_name(base[m], 'q')); });
    }
    for (auto &t : rd) t.join();
  }
  for (int m = 0; m < nm; m++) {
    uint8_t mg[8];
    if (R[m].read(mg, 8) != 8 || memcmp(mg, "scalce2", 7)) FAIL("%s is not a scalce archive\n", base[m].c_str());
    no_ac = 0;
    if (mg[6] == '2' && mg[7] >= '2') R[m].read(&no_ac, 4);
    Q[m].read(mg, 8);
    Nn[m].read(mg, 8);
    R[m].read(&len[m], 4);
    Q[m].read(&phred[m], 8);
  }
  uint8_t names = 0;
  std::string library = o.library;
  if (o.use_names) {  // decompress.cpp:219-237
    for (int m = 0; m < nm; m++) Nn[m].read(&names, 1);
    if (!names)
      for (int m = 0; m < nm; m++) {
        int64_t idx;
        Nn[m].read(&idx, 8);
        library.assign((const char *)Nn[m].v.data() + Nn[m].pos, Nn[m].v.size() - Nn[m].pos);
      }
  }
  // A mate's text comes down and is written by a thread of its own while the next mate is read, decoded and turned into text:
  // writing 63 GB of FASTQ per mate (200 M pairs x 150 bp) is most of the run, and the two mates are two files.
  std::vector<std::thread> writers;
  // a file's bytes are dropped as soon as the device has them, by a thread of its own: returning the 54 GB of a 200 M-pair
  // archive to the system took 4.5 s at the end of the run
  std::vector<std::thread> droppers;
  auto release = [&droppers](std::vector<uint8_t> &v) {
    if (v.size() < (64u << 20)) return;
    droppers.emplace_back([old = std::move(v)]() mutable { std::vector<uint8_t>().swap(old); });
    v.clear();
  };
  const double t_files = now() - t0;
  double t_decode = 0, t_records = 0;
  std::atomic<double> t_write{0};
  for (int m = 0; m < nm; m++) {
    const int L = len[m];
    const double ta = now();
    // qualities: arithmetic decoder on the device, or the raw q - offset bytes of a -A archive
    void *d_q = nullptr;
    uint64_t total = 0;
    if (!no_ac) {  // table + total + blocks -> GPU decoder
      std::vector<uint32_t> table(512000);
      if (Q[m].read(table.data(), 512000 * 4) != 512000 * 4) FAIL("truncated quality table\n");
      Q[m].read(&total, 8);
      const size_t nb = Q[m].v.size() - Q[m].pos;
      void *d_in = nullptr;
      HIPOK(hipMalloc(&d_in, nb + 64));
      HIPOK(hipMalloc(&d_q, total + 64));
      HIPOK(hipMemcpy(d_in, Q[m].v.data() + Q[m].pos, nb, hipMemcpyHostToDevice));
      release(Q[m].v);  // (gigabytes go back to the system beside the decoder, not behind the run)
      SCOK(ctx, scalce_ac_decode(ctx, table.data(), (const uint8_t *)d_in, nb, total, (uint8_t *)d_q, nullptr));
      hipFree(d_in);
    } else {
      total = Q[m].v.size() - Q[m].pos;
      HIPOK(hipMalloc(&d_q, total + 64));
      HIPOK(hipMemcpy(d_q, Q[m].v.data() + Q[m].pos, total, hipMemcpyHostToDevice));
    }
    const uint64_t nrec = L ? total / (uint64_t)L : 0;
    // records -> FASTQ text on the device (decompress.cpp:240-366).  Mate 1's bucket directory gives the core of every
    // record; mate-2 records carry no core (the reference lets `corlen` of the LAST mate-1 bucket leak into the mate-2
    // pass, decompress.cpp:250,269,332 -- not reproduced).
    const uint8_t *npay = names ? Nn[m].v.data() + Nn[m].pos : nullptr;
    const uint64_t nbytes_names = names ? Nn[m].v.size() - Nn[m].pos : 0;
    if (names && nbytes_names < nrec) FAIL("truncated name stream\n");
    const uint64_t cap = scalce_fastq_text_bytes(L, nrec, nbytes_names, names ? nullptr : library.c_str());
    void *d_text = nullptr;
    HIPOK(hipMalloc(&d_text, cap + 64));
    const double tb = now();
    t_decode += tb - ta;
    uint64_t text_bytes = 0;
    std::vector<uint64_t> roff;
    if (o.split) roff.resize((size_t)nrec + 1);
    SCOK(ctx, scalce_fastq_records(ctx, L, m == 0, R[m].v.data() + R[m].pos, R[m].v.size() - R[m].pos, nrec, (const uint8_t *)d_q,
                                   phred[m], npay, nbytes_names, library.c_str(), o.paired ? '1' + m : 0, (uint8_t *)d_text, cap,
                                   &text_bytes, o.split ? roff.data() : nullptr, nullptr));
    hipFree(d_q);
    release(R[m].v);
    release(Nn[m].v);
    t_records += now() - tb;
    // the text comes down in slices through pinned buffers while the previous slices are being written
    auto write_out = [&o, &t_write, m, nrec, text_bytes, d_text, roff = std::move(roff)]() {
      const double tw = now();
      HIPOK(hipSetDevice(0));
      Downloader down;
      char fn[4096];
      int part = 1;
      auto part_name = [&](int F) {
        if (o.out == "-") snprintf(fn, sizeof fn, "-");
        else if (o.split) snprintf(fn, sizeof fn, "%s.%d_%d.fastq", o.out.c_str(), part, F + 1);
        else snprintf(fn, sizeof fn, "%s_%d.fastq", o.out.c_str(), F + 1);
      };
      const uint64_t per = o.split ? (uint64_t)o.split : (nrec ? nrec : 1);
      for (uint64_t k0 = 0; k0 < nrec || k0 == 0; k0 += per, part++) {  // decompress.cpp:276-287: a new file every -S reads
        const uint64_t k1 = std::min<uint64_t>(nrec, k0 + per);
        const uint64_t b0 = o.split ? roff[(size_t)k0] : 0, b1 = o.split ? roff[(size_t)k1] : text_bytes;
        OutFile fo;
        part_name(m);
        fo.open(fn, false);
        down.range_to_file(static_cast<const uint8_t *>(d_text) + b0, b1 - b0, fo);
        fo.close();
        LOG("Created %s with %lld reads\n", fn, (long long)(k1 - k0));
        if (!nrec) break;
      }
      hipFree(d_text);
      const double dt = now() - tw;
      for (double cur = t_write.load(); !t_write.compare_exchange_weak(cur, cur + dt);) {}
    };
    if (nm == 2 && o.out != "-") writers.emplace_back(std::move(write_out));  // (stdout takes one mate: decompress.cpp refuses -r with "-")
    else write_out();
  }
  for (auto &t : writers) t.join();
  for (auto &t : droppers) t.join();
  LOG("\tTime elapsed: %.2f s (archive files read %.2f; qualities up and decoded %.2f; records to text %.2f; text down and written %.2f%s)\n",
      now() - t0, t_files, t_decode, t_records, t_write.load(), nm == 2 ? ", a thread per mate beside the next mate's decode" : "");
  return 0;
}

int main(int argc, char **argv) {
  const double t_main = now();
  Options o;
  LOG("SCALCE %s [MI355X / HIP]\n", SCALCE_VERSION);
  static struct option long_opt[] = {{"help", 0, 0, 'h'}, {"lossy-percentage", 1, 0, 'p'}, {"decompress", 0, 0, 'd'},
                                     {"compression", 1, 0, 'c'}, {"output", 1, 0, 'o'}, {"sample-size", 1, 0, 's'},
                                     {"no-qualities", 0, 0, 'Q'}, {"patterns", 1, 0, 'P'}, {"temp-directory", 1, 0, 't'},
                                     {"bucket-set-size", 1, 0, 'B'}, {"paired-end", 0, 0, 'r'}, {"skip-names", 1, 0, 'n'},
                                     {"split-reads", 1, 0, 'S'}, {"fasta", 0, 0, 'f'}, {"threads", 1, 0, 'T'},
                                     {"version", 0, 0, 'v'}, {"no-arithmetic", 0, 0, 'A'}, {"patterns-bin", 1, 0, 1000},
                                     {"gpus", 1, 0, 1001}, {0, 0, 0, 0}};
  int opt;
  while ((opt = getopt_long(argc, argv, "vhp:T:dc:o:fs:t:B:rQAn:P:S:", long_opt, 0)) != -1) {
    switch (opt) {
      case 'v': return 0;
      case 'h': fputs(HELP_TEXT, stdout); return 0;
      case 'A': o.no_ac = true; break;
      case 'f': case 'Q': FAIL("FASTA / no-quality mode is outside this build's scope (SURVEY.md section 2, #23)\n");
      case 'c':
        if (!strcmp(optarg, "gz") || !strcmp(optarg, "pigz")) o.container = 1;
        else if (!strcmp(optarg, "no")) o.container = 0;
        else if (!strcmp(optarg, "bz")) FAIL("bzip2 containers are outside this build's scope (SURVEY.md section 2, #18: host-side container code); use gz or no\n");
        else FAIL("Unknown compression mode. See help for details.\n");
        break;
      case 'B': {
        std::string s = optarg;
        const char al = s.empty() ? 0 : s.back();
        uint64_t unit = 1024 * 1024ull;
        if (al == 'G') unit *= 1024; else if (al != 'M') FAIL("Size parameter must be ended with G or M.\n");
        s.pop_back();
        o.bucket_set_size = unit * (uint64_t)atoi(s.c_str());
      } break;
      case 'p': o.lossy = atoi(optarg); break;
      case 'T': o.threads = atoi(optarg); break;
      case 'S': o.split = atoi(optarg); break;
      case 'r': o.paired = true; break;
      case 's': o.sample = atoi(optarg); break;
      case 'd': o.decompress = true; break;
      case 'P': o.patterns = optarg; break;
      case 't': o.temp = optarg; break;
      case 'o': o.out = optarg; break;
      case 'n': o.use_names = false; o.library = optarg; break;
      case 1000: o.patterns_bin = optarg; break;
      case 1001: o.gpus = atoi(optarg); break;
      default: fputs(HELP_TEXT, stdout); return 0;
    }
  }
  std::vector<std::string> files(argv + optind, argv + argc);
  // check_arguments, main.cpp:120-164
  if (o.out.empty()) FAIL("No output file specified.\n");
  if (!o.use_names && o.library.empty()) FAIL("No library name specified.\n");
  if (o.decompress && files.size() > 1) FAIL("Too many files specified (decompression only supports one file).\n");
  if (o.lossy < 0 || o.lossy > 100) FAIL("Percentage must be in range [0,100].\n");
  if (o.out == "-" && (o.split || o.paired)) FAIL("stdout can be only used with single-end file decompression. It cannot be used with --split-reads option!\n");
  if (files.empty()) FAIL("No input file specified.\n");
  for (auto &f : files) {
    struct stat st;
    if (stat(f.c_str(), &st) != 0) FAIL("File %s does not exist or it is not accessible.\n", f.c_str());
    if (o.paired) {
      std::string f2;
      if (!second_file(f, f2)) FAIL("Cannot get file name for paired end for file %s. File should contain character 1.\n", f.c_str());
      if (stat(f2.c_str(), &st) != 0) FAIL("File %s does not exist or it is not accessible.\n", f2.c_str());
    }
  }
  if (o.gpus > 1 && !o.decompress) {  // forks before anything touches a GPU
    // (a run that -B does not cut anywhere is one chunk: scalce_sharded_compress sends all its rows to rank 0)
    return multi_gpu_compress(o, files, argv[0]);
  }
  scalce_ctx *ctx = nullptr;
  if (scalce_ctx_create(0, &ctx)) FAIL("%s\n", scalce_last_error(ctx));
  bool is_text = false;
  std::vector<uint8_t> table = load_core_table(o, argv[0], is_text);
  if (is_text) SCOK(ctx, scalce_patterns_load_text(ctx, (const char *)table.data(), table.size()));
  else SCOK(ctx, scalce_patterns_load_bin(ctx, table.data(), table.size()));
  {
    const int hw = (int)std::thread::hardware_concurrency();
    g_threads = o.threads > 0 ? o.threads : std::max(1, std::min(64, hw - 1));
  }
  const double t_ready = now();
  const int rc = o.decompress ? do_decompress(o, files[0], ctx) : do_compress(o, files, ctx);
  scalce_ctx_destroy(ctx);
  LOG("\tProcess: device and core table ready %.2f s after main() began, %.2f s in all\n", t_ready - t_main, now() - t_main);
  LOG("Done!\n");
  return rc;
}
