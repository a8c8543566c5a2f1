/*
 * ref_full.cpp -- the WHOLE reference pipeline, run from its own objects (test infrastructure).
 *
 * oracle/Makefile compiles const, names, reads, qualities, arithmetic, buffio, compress and
 * decompress .cpp where they lie under /root/reference (bzlib.h is the real header of this image,
 * /opt/conda/include, taken with -idirafter so the system zlib.h stays in front) and links them with
 * this file.  Only main.cpp is left out (it includes sys/sysctl.h, which glibc 2.35 dropped): this
 * file defines the option globals main.cpp:62-80 defines, fills them from its own flag parser and
 * calls the reference's compress() (compress.h:45) / decompress() (decompress.h:45).  Every byte
 * of the archives and of the decompressed FASTQ is therefore written by the reference's own code.
 *
 * usage (argument layout of orc_cli, so one flag list drives both):
 *   ref_full {compress|decompress} <patterns.bin|-P list.txt> <input[,input...]> <out>
 *            [-r] [-A] [-p N] [-n lib] [-c gz|no|bz] [-B bytes] [-s N] [-T n] [-t tmpdir] [-S n]
 * The embedded core table is tests/golden/patterns.bin (attached like the reference Makefile:38-39
 * attaches its own); a patterns.bin argument is compared with it byte for byte and refused when
 * it differs, so a test can never believe it ran another table.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <string>
#include <vector>

#include "buffio.h"
#include "compress.h"
#include "const.h"
#include "decompress.h"

/* option globals of main.cpp:62-80 */
int _quality_sample_lines = 100000;
int _quality_lossy_percentage = 0;
char _use_second_file = 0;
char _is_fasta = 0;
char _use_names = 1;
uint64_t _file_buffer_size = 128 * 1024 * 1024;
uint64_t _max_bucket_set_size = 4ull * 1024 * 1024 * 1024;
char _temp_directory[MAXLINE] = "__temp__";
char _output_path[MAXLINE] = "";
char _library_name[MAXLINE] = "";
char _pattern_path[MAXLINE];
int _split_reads = 0;
int _compression_mode = IO_GZIP;
char _interleave = 0;
int64_t _time_elapsed = 0;
int _thread_count = 1;
int _decompress = 0;
int _no_ac = 0;
int _compress_qualities = 1;

extern char _binary_patterns_bin_start, _binary_patterns_bin_end; /* reads.cpp:327-328 */

static double now() {
  struct timeval t;
  gettimeofday(&t, 0);
  return t.tv_sec + 1e-6 * t.tv_usec;
}

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: ref_full {compress|decompress} <patterns.bin|-P list.txt> <input[,input...]> <out> [flags]\n");
    return 2;
  }
  int dec = !strcmp(argv[1], "decompress");
  int ai = 2;
  _pattern_path[0] = 0;
  if (!strcmp(argv[ai], "-P")) {
    strncpy(_pattern_path, argv[ai + 1], MAXLINE - 1);
    ai += 2;
  } else {
    FILE *f = fopen(argv[ai], "rb");
    if (!f) { perror(argv[ai]); return 2; }
    std::vector<char> b(&_binary_patterns_bin_end - &_binary_patterns_bin_start + 1);
    size_t n = fread(b.data(), 1, b.size(), f);
    fclose(f);
    if (n != b.size() - 1 || memcmp(b.data(), &_binary_patterns_bin_start, n)) {
      fprintf(stderr, "ref_full: %s is not the core table this binary embeds\n", argv[ai]);
      return 2;
    }
    ai += 1;
  }
  std::string inputs = argv[ai++];
  strncpy(_output_path, argv[ai++], MAXLINE - 1);
  _compression_mode = IO_SYS;
  for (; ai < argc; ai++) {
    const char *a = argv[ai];
    const char *v = ai + 1 < argc ? argv[ai + 1] : 0;
    if (!strcmp(a, "-r")) _use_second_file = 1;
    else if (!strcmp(a, "-A")) _no_ac = 1;
    else if (!strcmp(a, "-p") && v) { _quality_lossy_percentage = atoi(v); ai++; }
    else if (!strcmp(a, "-s") && v) { _quality_sample_lines = atoi(v); ai++; }
    else if (!strcmp(a, "-T") && v) { _thread_count = atoi(v); ai++; }
    else if (!strcmp(a, "-S") && v) { _split_reads = atoi(v); ai++; }
    else if (!strcmp(a, "-B") && v) { _max_bucket_set_size = strtoull(v, 0, 10); ai++; }
    else if (!strcmp(a, "-t") && v) { strncpy(_temp_directory, v, MAXLINE - 1); ai++; }
    else if (!strcmp(a, "-n") && v) { _use_names = 0; strncpy(_library_name, v, MAXLINE - 1); ai++; }
    else if (!strcmp(a, "-c") && v) {
      _compression_mode = !strcmp(v, "gz") ? IO_GZIP : !strcmp(v, "bz") ? IO_BZIP : IO_SYS;
      ai++;
    } else { fprintf(stderr, "ref_full: unknown option %s\n", a); return 2; }
  }
  std::vector<std::string> names;
  for (size_t p = 0; p <= inputs.size();) {
    size_t q = inputs.find(',', p);
    if (q == std::string::npos) q = inputs.size();
    names.push_back(inputs.substr(p, q - p));
    p = q + 1;
  }
  std::vector<char *> files;
  for (auto &s : names) files.push_back(&s[0]);
  _time_elapsed = TIME;
  double t0 = now();
  if (dec) {
    _decompress = 1;
    decompress(files[0], _output_path);
  } else {
    mkdir(_temp_directory, 0777); /* check_arguments does this, main.cpp:143-148; compress() removes it */
    compress(files.data(), (int)files.size(), _output_path, _pattern_path);
  }
  fprintf(stderr, "ref_full: %s %.3f s\n", argv[1], now() - t0);
  return 0;
}
