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
#include <sys/stat.h>
#include <sys/time.h>
#include <zlib.h>
#include <atomic>
#include <thread>
#include <algorithm>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/scalce_hip.h"

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
  int container = 1;                      // 0 plain, 1 gzip (main.cpp:181-184 at -T 1)
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
    "  -T, --threads INT           host threads that deflate the gz containers (default: cores - 1, at most 16);\n"
    "                              the hot path itself runs on the GPU\n"
    "  -t, --temp-directory STR    accepted for compatibility (nothing is spilled)\n"
    "  -S, --split-reads INT       decompression: reads per output part\n"
    "  -d, --decompress    -v, --version    -h, --help\n"
    "core table: --patterns-bin FILE or $SCALCE_PATTERNS or patterns.bin next to the executable\n";

// ---- small I/O helpers ------------------------------------------------------------------------------
static std::vector<uint8_t> read_maybe_gz(const std::string &path) {  // IO_GZIP reader (compress.cpp:756)
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) FAIL("Cannot read file %s\n", path.c_str());
  gzbuffer(f, 1 << 20);
  std::vector<uint8_t> out;
  struct stat st;
  if (stat(path.c_str(), &st) == 0) out.reserve((size_t)st.st_size + 64);
  std::vector<uint8_t> chunk(8 << 20);
  for (;;) {
    int k = gzread(f, chunk.data(), (unsigned)chunk.size());
    if (k < 0) FAIL("Read error on %s\n", path.c_str());
    if (k == 0) break;
    out.insert(out.end(), chunk.begin(), chunk.begin() + k);
  }
  gzclose(f);
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
struct OutFile {
  bool gz = false;
  FILE *f = nullptr;
  std::vector<uint8_t> pending;  // gz only
  void open(const std::string &path, bool gz_) {
    gz = gz_;
    f = path == "-" ? stdout : fopen(path.c_str(), "wb");
    if (!f) FAIL("Cannot create %s\n", path.c_str());
  }
  void raw(const uint8_t *b, size_t n) {
    while (n) {
      size_t k = n > (1u << 30) ? (1u << 30) : n;
      if (fwrite(b, 1, k, f) != k) FAIL("write failed\n");
      b += k; n -= k;
    }
  }
  void write(const void *p, size_t n) {
    const uint8_t *b = static_cast<const uint8_t *>(p);
    if (gz) pending.insert(pending.end(), b, b + n);
    else raw(b, n);
  }
  static void deflate_member(const uint8_t *src, size_t n, std::vector<uint8_t> &out) {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) FAIL("deflateInit2 failed\n");
    out.resize(deflateBound(&z, (uLong)n) + 64);
    z.next_in = const_cast<Bytef *>(src); z.avail_in = (uInt)n;
    z.next_out = out.data(); z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) FAIL("deflate failed\n");
    out.resize(z.total_out);
    deflateEnd(&z);
  }
  void close() {
    if (gz && f) {
      const size_t chunk = 4u << 20, nchunks = pending.empty() ? 1 : (pending.size() + chunk - 1) / chunk;
      std::vector<std::vector<uint8_t>> members(nchunks);
      std::atomic<size_t> next{0};
      auto work = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < nchunks;) {
          const size_t a = i * chunk, b = std::min(pending.size(), a + chunk);
          deflate_member(pending.data() + a, b - a, members[i]);
        }
      };
      const int nt = (int)std::min<size_t>((size_t)std::max(1, g_threads), nchunks);
      std::vector<std::thread> pool;
      for (int t = 1; t < nt; t++) pool.emplace_back(work);
      work();
      for (auto &t : pool) t.join();
      for (auto &m : members) raw(m.data(), m.size());
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

// sampling loop of quality_mapping_init (qualities.cpp:64-97) on the in-memory text
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
  std::vector<uint8_t> text[2];
  uint64_t original = 0;
  int32_t qhist[128];
  scalce_params p;
  scalce_params_default(&p);
  p.paired = o.paired; p.use_names = o.use_names; p.no_ac = o.no_ac; p.bucket_set_size = o.bucket_set_size;
  LOG("Preprocessing FASTQ files ...\n");
  for (size_t F = 0; F < files.size(); F++)
    for (int m = 0; m < nm; m++) {
      std::string path = files[F];
      if (m && !second_file(files[F], path))
        FAIL("Cannot get file name for paired end for file %s. File should contain character 1.\n", files[F].c_str());
      std::vector<uint8_t> t = read_maybe_gz(path);
      struct stat st;
      if (stat(path.c_str(), &st) == 0) original += (uint64_t)st.st_size;
      if (F == 0) {  // get_quality_stats looks at the first file only (compress.cpp:761)
        int rl = 0;
        sample_stats(t, o.sample, qhist, rl);
        scalce_qmap_init(&p.qmap[m], qhist, o.lossy);
        p.read_len[m] = rl;
        LOG("\tPaired end #%d, quality offset: %d\n\t               read length: %d\n", m + 1, p.qmap[m].offset, rl);
      }
      if (text[m].empty()) text[m].swap(t); else text[m].insert(text[m].end(), t.begin(), t.end());
    }
  if (p.read_len[0] <= 0) FAIL("Cannot determine the read length of %s\n", files[0].c_str());
  const uint64_t max_text = (text[0].size() > text[1].size() ? text[0].size() : text[1].size()) + 64;
  const uint64_t max_reads = text[0].size() / (2 * (uint64_t)p.read_len[0] + 4) + 16;
  scalce_batch *b = nullptr;
  SCOK(ctx, scalce_batch_create(ctx, &p, max_reads, max_text, &b));
  void *d[2] = {nullptr, nullptr};
  for (int m = 0; m < nm; m++) {
    HIPOK(hipMalloc(&d[m], text[m].size() + 64));
    HIPOK(hipMemcpy(d[m], text[m].data(), text[m].size(), hipMemcpyHostToDevice));
  }
  const double t1 = now();
  SCOK(ctx, scalce_batch_compress(b, (const uint8_t *)d[0], text[0].size(), (const uint8_t *)d[1], text[1].size(), nullptr));
  SCOK(ctx, scalce_batch_finish(b, nullptr));
  const double t2 = now();
  const uint64_t N = scalce_batch_reads(b);
  LOG("\tDone with file %s, %llu reads found\n", files[0].c_str(), (unsigned long long)N);
  uint32_t st4[5] = {0, 0, 0, 0, 0};
  scalce_batch_stats(b, st4);

  // final writer: headers of combine_and_compress_with_split (compress.cpp:263-343)
  const uint8_t magic[8] = {'s', 'c', 'a', 'l', 'c', 'e', '2', '2'};
  const bool gz = o.container == 1;
  uint64_t new_size = 0;
  std::vector<uint8_t> names = o.use_names ? fetch(ctx, b, SCALCE_OUT_NAMES, 0) : std::vector<uint8_t>();
  for (int m = 0; m < nm; m++) {
    char fn[4096];
    OutFile fR, fQ, fN;
    snprintf(fn, sizeof fn, "%s_%d.scalcer", o.out.c_str(), m + 1); fR.open(fn, gz);
    const int32_t noac = o.no_ac, len32 = p.read_len[m];
    fR.write(magic, 8); fR.write(&noac, 4); fR.write(&len32, 4);
    { auto v = fetch(ctx, b, SCALCE_OUT_READS, m); fR.write(v.data(), v.size()); }
    fR.close();
    snprintf(fn, sizeof fn, "%s_%d.scalceq", o.out.c_str(), m + 1); fQ.open(fn, o.no_ac ? gz : false);  // :249
    const int64_t phred = p.qmap[0].offset;  // mate 1's offset for both (compress.cpp:294,816-817)
    fQ.write(magic, 8); fQ.write(&phred, 8);
    if (!o.no_ac) {
      auto tb = fetch(ctx, b, SCALCE_OUT_TABLE, m);
      fQ.write(tb.data(), tb.size());
      const uint64_t total = N * (uint64_t)p.read_len[m];
      fQ.write(&total, 8);
    }
    { auto v = fetch(ctx, b, SCALCE_OUT_QUAL, m); fQ.write(v.data(), v.size()); }
    fQ.close();
    snprintf(fn, sizeof fn, "%s_%d.scalcen", o.out.c_str(), m + 1); fN.open(fn, gz);
    const uint8_t un = o.use_names ? 1 : 0;
    fN.write(magic, 8); fN.write(&un, 1);
    if (o.use_names) fN.write(names.data(), names.size());  // mate 2 repeats mate 1's names (:450-454)
    else { const int64_t z = 0; fN.write(&z, 8); fN.write(o.library.data(), o.library.size()); }
    fN.close();
    for (const char *ext : {"r", "q", "n"}) {
      snprintf(fn, sizeof fn, "%s_%d.scalce%s", o.out.c_str(), m + 1, ext);
      struct stat st;
      if (stat(fn, &st) == 0) new_size += (uint64_t)st.st_size;
    }
  }
  const void *dc = nullptr;
  uint64_t nc = 0;
  SCOK(ctx, scalce_batch_output(b, SCALCE_OUT_BUCKET_COUNTS, 0, &dc, &nc));
  uint64_t unbucketed = 0;
  if (nc >= 8) SCOK(ctx, scalce_memcpy_d2h(ctx, &unbucketed, (const uint8_t *)dc + nc - 8, 8));
  scalce_batch_destroy(b);
  for (int m = 0; m < nm; m++) hipFree(d[m]);
  const double t3 = now();
  LOG("Statistics:\n\tTotal number of reads: %llu\n\tRead length: first end %d\n", (unsigned long long)N, p.read_len[0]);
  if (o.paired) LOG("\t             second end %d\n", p.read_len[1]);
  LOG("\tUnbucketed reads count: %llu, bucketed percentage %.2lf\n", (unsigned long long)unbucketed,
      N ? 100.0 * (double)(N - unbucketed) / (double)N : 0.0);
  LOG("\tLossy percentage: %d\n", o.lossy);
  LOG("\tTie reads: %u, fixed-point sweeps: %u, spill chunks: %u\n", st4[0], st4[2], st4[3]);
  LOG("\tTime elapsed: %.2f s (read+upload %.2f, GPU hot path %.2f, write %.2f)\n", t3 - t0, t1 - t0, t2 - t1, t3 - t2);
  LOG("\tOriginal size: %.2lfM, new size: %.2lfM, compression factor: %.2lf\n", original / (1024.0 * 1024.0),
      new_size / (1024.0 * 1024.0), new_size ? (double)original / (double)new_size : 0.0);
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
  for (int m = 0; m < nm; m++) {
    R[m].v = read_maybe_gz(scalce_name(base[m], 'r'));  // container sniffing: zlib reads plain and gzip alike
    Nn[m].v = read_maybe_gz(scalce_name(base[m], 'n'));
    Q[m].v = read_maybe_gz(scalce_name(base[m], 'q'));
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
  for (int m = 0; m < nm; m++) {
    const int L = len[m];
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
    uint64_t text_bytes = 0;
    std::vector<uint64_t> roff;
    if (o.split) roff.resize((size_t)nrec + 1);
    SCOK(ctx, scalce_fastq_records(ctx, L, m == 0, R[m].v.data() + R[m].pos, R[m].v.size() - R[m].pos, nrec, (const uint8_t *)d_q,
                                   phred[m], npay, nbytes_names, library.c_str(), o.paired ? '1' + m : 0, (uint8_t *)d_text, cap,
                                   &text_bytes, o.split ? roff.data() : nullptr, nullptr));
    hipFree(d_q);
    std::vector<char> text((size_t)text_bytes);
    HIPOK(hipMemcpy(text.data(), d_text, text_bytes, hipMemcpyDeviceToHost));
    hipFree(d_text);
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
      fo.write(text.data() + b0, (size_t)(b1 - b0));
      fo.close();
      LOG("Created %s with %lld reads\n", fn, (long long)(k1 - k0));
      if (!nrec) break;
    }
  }
  LOG("\tTime elapsed: %.2f s\n", now() - t0);
  return 0;
}

int main(int argc, char **argv) {
  Options o;
  LOG("SCALCE %s [MI355X / HIP]\n", SCALCE_VERSION);
  static struct option long_opt[] = {{"help", 0, 0, 'h'}, {"lossy-percentage", 1, 0, 'p'}, {"decompress", 0, 0, 'd'},
                                     {"compression", 1, 0, 'c'}, {"output", 1, 0, 'o'}, {"sample-size", 1, 0, 's'},
                                     {"no-qualities", 0, 0, 'Q'}, {"patterns", 1, 0, 'P'}, {"temp-directory", 1, 0, 't'},
                                     {"bucket-set-size", 1, 0, 'B'}, {"paired-end", 0, 0, 'r'}, {"skip-names", 1, 0, 'n'},
                                     {"split-reads", 1, 0, 'S'}, {"fasta", 0, 0, 'f'}, {"threads", 1, 0, 'T'},
                                     {"version", 0, 0, 'v'}, {"no-arithmetic", 0, 0, 'A'}, {"patterns-bin", 1, 0, 1000},
                                     {0, 0, 0, 0}};
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
        else if (!strcmp(optarg, "bz")) FAIL("bzip2 containers are not built (no bzlib in this image); use gz or no\n");
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
  scalce_ctx *ctx = nullptr;
  if (scalce_ctx_create(0, &ctx)) FAIL("%s\n", scalce_last_error(ctx));
  bool is_text = false;
  std::vector<uint8_t> table = load_core_table(o, argv[0], is_text);
  if (is_text) SCOK(ctx, scalce_patterns_load_text(ctx, (const char *)table.data(), table.size()));
  else SCOK(ctx, scalce_patterns_load_bin(ctx, table.data(), table.size()));
  {
    const int hw = (int)std::thread::hardware_concurrency();
    g_threads = o.threads > 0 ? o.threads : std::max(1, std::min(16, hw - 1));
  }
  const int rc = o.decompress ? do_decompress(o, files[0], ctx) : do_compress(o, files, ctx);
  scalce_ctx_destroy(ctx);
  LOG("Done!\n");
  return rc;
}
