/* fastq_digest.c -- order-independent digest of the records of a FASTQ file (or of the PAIRS of two files read in
 * step): count, sum and xor of a 64-bit hash per record.  Two files hold the same multiset of records (pairs) iff --
 * up to hash collisions -- their digests agree; used to check the decompressed output of full-size runs against the
 * input, where sorting hundreds of millions of records is not an option.
 * usage: fastq_digest FILE_1 [FILE_2]      build: gcc -O2 -msse4.2 -o fastq_digest fastq_digest.c */
#include <nmmintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { FILE *f; char *buf; size_t cap, len, pos; } Rd;
static void rd_open(Rd *r, const char *p) {
  r->f = fopen(p, "rb");
  if (!r->f) { perror(p); exit(2); }
  r->cap = 64u << 20; r->buf = (char *)malloc(r->cap); r->len = r->pos = 0;
}
/* next line incl. its newline; NULL at the end */
static char *rd_line(Rd *r, size_t *n) {
  for (;;) {
    char *nl = r->pos < r->len ? (char *)memchr(r->buf + r->pos, '\n', r->len - r->pos) : NULL;
    if (nl) { char *s = r->buf + r->pos; *n = (size_t)(nl - s) + 1; r->pos += *n; return s; }
    memmove(r->buf, r->buf + r->pos, r->len - r->pos);
    r->len -= r->pos; r->pos = 0;
    if (r->len == r->cap) { fprintf(stderr, "line too long\n"); exit(2); }
    size_t k = fread(r->buf + r->len, 1, r->cap - r->len, r->f);
    if (!k) { if (r->len) { fprintf(stderr, "no trailing newline\n"); exit(2); } return NULL; }
    r->len += k;
  }
}
static uint64_t mix(uint64_t h, const char *s, size_t n) {
  uint64_t a = h, b = ~h;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, s + i, 8); a = _mm_crc32_u64(a, v); b = _mm_crc32_u64(b, v ^ 0x9E3779B97F4A7C15ull); }
  for (; i < n; i++) { a = _mm_crc32_u8((uint32_t)a, (uint8_t)s[i]); b = _mm_crc32_u8((uint32_t)b, (uint8_t)(s[i] ^ 0x5A)); }
  return (a << 32) ^ b ^ (n * 0xD6E8FEB86659FD93ull);
}
int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: fastq_digest FILE_1 [FILE_2]\n"); return 2; }
  Rd r[2];
  int nf = argc > 2 ? 2 : 1;
  for (int i = 0; i < nf; i++) rd_open(&r[i], argv[1 + i]);
  uint64_t cnt = 0, sum = 0, x = 0;
  for (;;) {
    uint64_t h = 0x243F6A8885A308D3ull;
    int end = 0;
    for (int i = 0; i < nf && !end; i++)
      for (int l = 0; l < 4; l++) {
        size_t n;
        char *s = rd_line(&r[i], &n);
        if (!s) { if (l || i) { fprintf(stderr, "truncated record / mates of different length\n"); return 2; } end = 1; break; }
        if (l == 2) { s = "+\n"; n = 2; }  /* the '+' line may repeat the name: canonical form */
        h = mix(h, s, n);
      }
    if (end) break;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    cnt++; sum += h; x ^= h;
  }
  printf("%llu records%s sum %016llx xor %016llx\n", (unsigned long long)cnt, nf == 2 ? " (pairs)" : "", (unsigned long long)sum, (unsigned long long)x);
  return 0;
}
