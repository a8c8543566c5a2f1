/*
 * ref_driver.cpp -- harness around the REAL reference objects (test infrastructure).
 *
 * oracle/Makefile compiles const.cpp, names.cpp, reads.cpp, qualities.cpp and
 * arithmetic.cpp straight from /root/reference (no copies, no shims) and links them
 * with this file.  The harness only plays the role of main()/thread(): it sets the
 * option globals that main.cpp defines, feeds records to the reference's own
 * aho_search / output_name / output_read / output_quality / aho_trie_bucket /
 * bin_prepare / ac_stat / ac_coder / ac_decoder, and dumps what they return.
 *
 * The whole-file side (quality_mapping_init's sampling loop, ac_write/ac_read framing,
 * bin_dump, compress(), decompress()) is ref_full.cpp's job; this harness looks INSIDE the
 * path: per-read tokens, order, records, counters, coder bytes.
 *
 * usage: ref_driver <fastq> <outdir> [-P patterns.txt] [-q qmap.txt] [-2 fastq2 [-q2 qmap2.txt]] [-f factor] [-n] [-t]
 *   -P  text core list (read_patterns_from_file) instead of the embedded patterns.bin
 *   -q  file with 129 integers: offset, values[0..127]  (default: offset 33, identity)
 *   -2  second mate (-r): every record also goes through output_read(read2, .., 0, 0) and
 *       output_quality(qual2, read2, qmap + 1, .., 1) exactly as thread() does (compress.cpp:692-699);
 *       packed2.bin / qual2.bin / freq4_2.u64 / ac2.bin are the mate-2 streams (same record order)
 *   -f  shrink factor of compress.cpp:297-313 applied to the counters before ac_stat is built
 *       (p = p / factor, 0 -> 1: those three lines live in the final writer, which needs buffio and
 *       is not linked, so they are repeated here); table.u32 / table2.u32 hold the scaled table
 *   -n  names off (-n lib)
 *   -t  timing mode (bench.py's cpu_baseline of kind "reference"): the compress path only -- no decoder check,
 *       nothing but ac.bin is written -- and the seconds spent in it go to stderr
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <string>
#include <vector>

#include "arithmetic.h"
#include "const.h"
#include "names.h"
#include "qualities.h"
#include "reads.h"

/* option globals normally defined in main.cpp:62-80 and compress.cpp:61-63 */
int _quality_sample_lines = 100000;
int _quality_lossy_percentage = 0;
char _use_second_file = 0;
char _is_fasta = 0;
char _use_names = 1;
uint64_t _file_buffer_size = 128 * 1024 * 1024;
uint64_t _max_bucket_set_size = 2ull * 1024 * 1024 * 1024;
char _temp_directory[MAXLINE] = "__temp__";
char _library_name[MAXLINE] = "";
char _pattern_path[MAXLINE];
int _split_reads = 0;
int _compression_mode = IO_SYS;
char _interleave = 0;
int64_t _time_elapsed = 0;
int _thread_count = 1;
int _decompress = 0;
int _no_ac = 0;
int _compress_qualities = 1;
/* read_length[2], reads_count: compress.cpp:61-62 (compress.o is linked, so nothing is left unresolved) */

void bin_prepare(aho_trie *t); /* reads.cpp:54 */
extern char _binary_patterns_bin_start, _binary_patterns_bin_end; /* reads.cpp:327-328 */

static void dump(const std::string &path, const void *p, size_t n) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(2); }
  fwrite(p, 1, n, f);
  fclose(f);
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: ref_driver <fastq> <outdir> [-P txt] [-q qmap] [-n]\n"); return 2; }
  std::string fq = argv[1], out = argv[2];
  const char *ptxt = 0, *qfile = 0, *qfile2 = 0, *fq2 = 0;
  int timing = 0, factor = 1;
  for (int i = 3; i < argc; i++) {
    if (!strcmp(argv[i], "-P") && i + 1 < argc) ptxt = argv[++i];
    else if (!strcmp(argv[i], "-q") && i + 1 < argc) qfile = argv[++i];
    else if (!strcmp(argv[i], "-q2") && i + 1 < argc) qfile2 = argv[++i];
    else if (!strcmp(argv[i], "-2") && i + 1 < argc) { fq2 = argv[++i]; _use_second_file = 1; }
    else if (!strcmp(argv[i], "-f") && i + 1 < argc) factor = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-n")) _use_names = 0;
    else if (!strcmp(argv[i], "-t")) timing = 1;
  }
  struct timespec ts0;
  clock_gettime(CLOCK_MONOTONIC, &ts0);
  quality_mapping qmaps[2];
  const char *qfiles[2] = {qfile, qfile2};
  for (int m = 0; m < 2; m++) {
    quality_mapping &q = qmaps[m];
    q.offset = 33;
    for (int i = 0; i < 128; i++) q.values[i] = i;
    if (qfiles[m]) {
      FILE *f = fopen(qfiles[m], "r");
      if (!f || fscanf(f, "%d", &q.offset) != 1) { fprintf(stderr, "bad qmap\n"); return 2; }
      for (int i = 0; i < 128; i++) if (fscanf(f, "%d", &q.values[i]) != 1) { fprintf(stderr, "bad qmap\n"); return 2; }
      fclose(f);
    }
  }
  quality_mapping &qm = qmaps[0];

  aho_trie *trie = ptxt ? read_patterns_from_file(ptxt) : read_patterns();

  FILE *f = fopen(fq.c_str(), "r");
  if (!f) { perror(fq.c_str()); return 2; }
  FILE *f2 = fq2 ? fopen(fq2, "r") : 0;
  if (fq2 && !f2) { perror(fq2); return 2; }
  static char name[MAXLINE], read[MAXLINE], plus[MAXLINE], qual[MAXLINE];
  static char name2[MAXLINE], read2[MAXLINE], plus2[MAXLINE], qual2[MAXLINE];
  std::vector<int32_t> tok;
  std::vector<uint8_t> packed, names, quals, packed2, quals2;
  int L2 = 0;
  std::vector<bin_node *> nodes_in_order;
  uint8_t outbuf[MAXLINE * 5];
  int64_t N = 0;
  int L = 0;
  while (fgets(name, MAXLINE, f) && fgets(read, MAXLINE, f) && fgets(plus, MAXLINE, f) && fgets(qual, MAXLINE, f)) {
    if (!L) { L = strlen(read) - 1; read_length[0] = L; }
    /* the body of thread(), compress.cpp:673-706, single mate */
    aho_trie *bucket;
    read_data rd;
    rd.data = outbuf;
    int n = aho_search(read, trie, &bucket);
    rd.sz = output_name(name, rd.data);
    names.insert(names.end(), rd.data, rd.data + rd.sz);
    int before = rd.sz;
    if (n != -1) {
      rd.sz += output_read(read, rd.data + rd.sz, n - bucket->level + 1, bucket->level);
      rd.end = n + 1;
    } else {
      rd.sz += output_read(read, rd.data + rd.sz, 0, 0);
      rd.end = 0;
    }
    packed.insert(packed.end(), rd.data + before, rd.data + rd.sz);
    before = rd.sz;
    rd.sz += output_quality(qual, read, &qm, rd.data + rd.sz, 0);
    quals.insert(quals.end(), rd.data + before, rd.data + rd.sz);
    rd.of = rd.sz;
    if (f2) { /* compress.cpp:692-699: the mate is never searched for a core */
      if (!(fgets(name2, MAXLINE, f2) && fgets(read2, MAXLINE, f2) && fgets(plus2, MAXLINE, f2) && fgets(qual2, MAXLINE, f2))) {
        fprintf(stderr, "mate 2 is shorter than mate 1\n");
        return 2;
      }
      if (!L2) { L2 = strlen(read2) - 1; read_length[1] = L2; }
      before = rd.sz;
      rd.sz += output_read(read2, rd.data + rd.sz, 0, 0);
      packed2.insert(packed2.end(), rd.data + before, rd.data + rd.sz);
      before = rd.sz;
      rd.sz += output_quality(qual2, read2, &qmaps[1], rd.data + rd.sz, 1);
      quals2.insert(quals2.end(), rd.data + before, rd.data + rd.sz);
    }
    rd.read_length = (int32_t)N; /* unused outside PACBIO builds: carries the input index */
    bin_node *bn = aho_trie_bucket(bucket, &rd);
    memcpy(bn->data.data, rd.data, rd.sz);
    tok.push_back(bucket->output);
    tok.push_back(rd.end);
    N++;
  }
  fclose(f);
  reads_count = N;

  if (!timing) {
    dump(out + "/tok.i32", tok.data(), tok.size() * 4);
    dump(out + "/packed.bin", packed.data(), packed.size());
    dump(out + "/names.bin", names.data(), names.size());
    dump(out + "/qual.bin", quals.data(), quals.size());
    dump(out + "/freq4.u64", ac_freq4[0], sizeof(uint64_t) * AC_DEPTH * AC_DEPTH * AC_DEPTH);
    if (f2) {
      dump(out + "/packed2.bin", packed2.data(), packed2.size());
      dump(out + "/qual2.bin", quals2.data(), quals2.size());
      dump(out + "/freq4_2.u64", ac_freq4[1], sizeof(uint64_t) * AC_DEPTH * AC_DEPTH * AC_DEPTH);
    }
  }
  if (f2) fclose(f2);

  /* pattern -> BFS id (reads.cpp:296): walk each core through the automaton */
  int np = 0;
  if (ptxt) {
    while (patterns[np]) np++; /* read_patterns_from_file zero-fills the array (reads.cpp:386) */
  } else {                     /* embedded blob: sum the group counts (reads.cpp:342-351) */
    const char *b = &_binary_patterns_bin_start, *e = &_binary_patterns_bin_end;
    while (b < e) {
      int16_t ln; int32_t cnt;
      memcpy(&ln, b, 2); memcpy(&cnt, b + 2, 4);
      b += 6 + (size_t)cnt * (ln / 4 + (ln % 4 != 0));
      np += cnt;
    }
  }
  std::vector<int32_t> ids(np);
  for (int p = 0; p < np; p++) {
    aho_trie *c = trie;
    for (const char *s = patterns[p]; *s && *s != '\n'; s++) c = c->child[getval(*s)];
    ids[p] = (c->output == p) ? c->id : -1;
  }
  if (!timing) dump(out + "/ids.i32", ids.data(), ids.size() * 4);

  /* emission order: the traversal of aho_output (reads.cpp:466-499) with bin_dump
   * replaced by a walk of the sorted list that bin_prepare leaves behind */
  std::vector<int64_t> order;
  {
    std::vector<aho_trie *> q;
    std::vector<char> seen(5000000 * 4, 0);
    seen[trie->id] = 1;
    for (int i = 0; i < 4; i++) { q.push_back(trie->child[i]); seen[trie->child[i]->id] = 1; }
    for (size_t h = 0; h < q.size(); h++) {
      aho_trie *cur = q[h];
      if (cur->bin.size) {
        bin_prepare(cur);
        for (bin_node *b = cur->bin.first; b; b = b->next) order.push_back(b->data.read_length);
      }
      for (int i = 0; i < 4; i++)
        if (cur->child[i] && !seen[cur->child[i]->id]) { q.push_back(cur->child[i]); seen[cur->child[i]->id] = 1; }
    }
    if (trie->bin.size) {
      bin_prepare(trie);
      for (bin_node *b = trie->bin.first; b; b = b->next) order.push_back(b->data.read_length);
    }
  }
  if (!timing) dump(out + "/order.i64", order.data(), order.size() * 8);

  /* arithmetic coder on the reordered quality stream(s); the table is scaled by `factor` first (1 when N*L < 2^32) */
  const size_t BS = 10 * 1024 * 1024;
  std::vector<uint8_t> enc, blk(BS * 2), dec(BS);
  size_t bad = 0;
  for (int m = 0; m < (f2 ? 2 : 1); m++) {
    const int Lm = m ? L2 : L;
    const std::vector<uint8_t> &qin = m ? quals2 : quals;
    std::vector<uint8_t> qs((size_t)N * Lm);
    for (int64_t k = 0; k < N; k++) memcpy(&qs[(size_t)k * Lm], &qin[(size_t)order[k] * Lm], Lm);
    static ac_stat as;
    std::vector<uint32_t> table(AC_DEPTH * AC_DEPTH * AC_DEPTH);
    for (int i = 0; i < AC_DEPTH * AC_DEPTH * AC_DEPTH; i++) {
      uint64_t *p = &ac_freq4[m][i];
      *p = *p / factor; /* compress.cpp:310-313 */
      if (*p == 0) *p = 1;
      table[i] = (uint32_t)*p;
    }
    if (!timing) dump(out + (m ? "/table2.u32" : "/table.u32"), table.data(), table.size() * 4);
    as = ac_stat(ac_freq3[m], ac_freq4[m]);
    enc.clear();
    for (size_t off = 0; off < qs.size(); off += BS) {
      size_t n = qs.size() - off < BS ? qs.size() - off : BS;
      ac_coder ax(blk.data(), &as);
      ax.write(&qs[off], (int)n);
      ax.flush();
      uint32_t sz = (uint32_t)(ax.output() - blk.data());
      enc.insert(enc.end(), (uint8_t *)&sz, (uint8_t *)&sz + 4);
      enc.insert(enc.end(), blk.data(), blk.data() + sz);
      if (timing) continue;
      ac_decoder ad(&as, blk.data());
      ad.read(dec.data(), (int)n);
      if (memcmp(dec.data(), &qs[off], n)) bad++;
    }
    dump(out + (m ? "/ac2.bin" : "/ac.bin"), enc.data(), enc.size());
  }
  if (timing) {
    struct timespec ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    fprintf(stderr, "ref_driver timing: %.3f s for the compress path of %lld reads (1 thread)\n",
            (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec), (long long)N);
  }
  fprintf(stderr, "ref_driver: %lld reads, L=%d, %d cores, %zu AC bytes, decode %s\n", (long long)N, L, np,
          enc.size(), bad ? "MISMATCH" : "ok");
  return bad ? 1 : 0;
}
