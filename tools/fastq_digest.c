/* fastq_digest.c -- order-independent digest of the records of a FASTQ file (or of the PAIRS of two files read in
 * step): count, sum and xor of a 64-bit hash per record.  Two files hold the same multiset of records (pairs) iff --
 * up to hash collisions -- their digests agree; used to check the decompressed output of full-size runs against the
 * input, where sorting hundreds of millions of records is not an option.
 * usage: fastq_digest FILE_1 [FILE_2]      build: gcc -O2 -msse4.2 -o fastq_digest fastq_digest.c
 *        fastq_digest --lossy MAPFILE FILE_1   (round 5) digest of what a LOSSY archive must decode to: MAPFILE holds the
 *        offset and the 128 values of the quality map (text: 129 integers, qualities.cpp:99-174); every record goes through
 *        q' = map[q] - offset, an upper-case 'N' forces q' = 0 (qualities.cpp:183), a base whose q' is 0 comes back as 'N',
 *        any other base as A C G T by getval (const.cpp:47-49: a c g t like A C G T, every other letter like A), the quality
 *        character as q' + offset (decompress.cpp:340-355) -- the transformation the reference's own round trip applies */
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
  int lossy = 0, off = 0, map[128];
  if (argc >= 4 && !strcmp(argv[1], "--lossy")) {
    FILE *m = fopen(argv[2], "r");
    if (!m || fscanf(m, "%d", &off) != 1) { fprintf(stderr, "cannot read the quality map\n"); return 2; }
    for (int i = 0; i < 128; i++) if (fscanf(m, "%d", &map[i]) != 1) { fprintf(stderr, "short quality map\n"); return 2; }
    fclose(m);
    lossy = 1; argv += 2; argc -= 2;
  }
  if (argc < 2) { fprintf(stderr, "usage: fastq_digest [--lossy MAPFILE] FILE_1 [FILE_2]\n"); return 2; }
  Rd r[2];
  int nf = argc > 2 ? 2 : 1;
  if (lossy && nf != 1) { fprintf(stderr, "--lossy takes one file\n"); return 2; }
  for (int i = 0; i < nf; i++) rd_open(&r[i], argv[1 + i]);
  static char bases[1 << 16], quals[1 << 16];
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
        if (lossy && l == 1) {               /* held back until the quality line says which bases come back as 'N' */
          if (n > sizeof bases) { fprintf(stderr, "line too long\n"); return 2; }
          memcpy(bases, s, n);
          continue;
        }
        if (lossy && l == 3) {
          for (size_t k = 0; k + 1 < n; k++) {
            int q = bases[k] == 'N' ? 0 : map[(unsigned char)s[k] & 127] - off;
            const char c = bases[k] & 0xDF;   /* getval: only C, G, T (either case) are not 0 */
            bases[k] = q == 0 ? 'N' : (c == 'C' || c == 'G' || c == 'T') ? c : 'A';
            quals[k] = (char)(q + off);
          }
          quals[n - 1] = '\n';
          h = mix(h, bases, n);
          h = mix(h, "+\n", 2);
          h = mix(h, quals, n);
          continue;
        }
        if (lossy && l == 2) continue;
        h = mix(h, s, n);
      }
    if (end) break;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    cnt++; sum += h; x ^= h;
  }
  printf("%llu records%s sum %016llx xor %016llx\n", (unsigned long long)cnt, nf == 2 ? " (pairs)" : "", (unsigned long long)sum, (unsigned long long)x);
  return 0;
}
