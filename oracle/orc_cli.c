/* orc_cli.c -- command line around the CPU oracle (test infrastructure; used by tests and by
 * bench.py's cpu_baseline leg).  Flag letters follow the reference CLI (main.cpp:187-296). */
#include "scalce_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

static uint8_t *slurp(const char *p, size_t *n) {
  FILE *f = fopen(p, "rb");
  if (!f) { perror(p); exit(2); }
  fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t *b = (uint8_t *)malloc(*n + 1);
  if (fread(b, 1, *n, f) != *n) { perror(p); exit(2); }
  fclose(f);
  return b;
}

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: orc_cli {compress|decompress} <patterns.bin|-P list.txt> <input> <out> "
                    "[-r] [-A] [-p N] [-n lib] [-c gz|no] [-B bytes] [-s N] [-T n]\n");
    return 2;
  }
  int decompress = !strcmp(argv[1], "decompress");
  int ai = 2;
  orc_trie *t;
  size_t n;
  if (!strcmp(argv[ai], "-P")) { uint8_t *b = slurp(argv[ai + 1], &n); t = orc_trie_from_text((char *)b, n); ai += 2; }
  else { uint8_t *b = slurp(argv[ai], &n); t = orc_trie_from_bin(b, n); ai += 1; }
  if (!t) { fprintf(stderr, "(ERROR) bad core table\n"); return 1; }
  const char *in = argv[ai++], *out = argv[ai++];
  orc_opts o;
  orc_opts_default(&o);
  o.gz = 0;
  for (; ai < argc; ai++) {
    if (!strcmp(argv[ai], "-r")) o.paired = 1;
    else if (!strcmp(argv[ai], "-A")) o.no_ac = 1;
    else if (!strcmp(argv[ai], "-v")) o.verbose = 1;
    else if (!strcmp(argv[ai], "-p") && ai + 1 < argc) o.lossy = atoi(argv[++ai]);
    else if (!strcmp(argv[ai], "-s") && ai + 1 < argc) o.sample = atoi(argv[++ai]);
    else if (!strcmp(argv[ai], "-T") && ai + 1 < argc) o.threads = atoi(argv[++ai]);
    else if (!strcmp(argv[ai], "-B") && ai + 1 < argc) o.bucket_set_size = strtoull(argv[++ai], 0, 10);
    else if (!strcmp(argv[ai], "-n") && ai + 1 < argc) { o.use_names = 0; o.library = argv[++ai]; }
    else if (!strcmp(argv[ai], "-c") && ai + 1 < argc) o.gz = !strcmp(argv[++ai], "gz");
    else { fprintf(stderr, "unknown option %s\n", argv[ai]); return 2; }
  }
  struct timeval a, b;
  gettimeofday(&a, 0);
  int rc = decompress ? orc_decompress_files(t, in, out, &o) : orc_compress_files(t, in, out, &o);
  gettimeofday(&b, 0);
  fprintf(stderr, "orc_cli: %s rc=%d %.3f s\n", argv[1], rc, (b.tv_sec - a.tv_sec) + 1e-6 * (b.tv_usec - a.tv_usec));
  orc_trie_free(t);
  return rc;
}
