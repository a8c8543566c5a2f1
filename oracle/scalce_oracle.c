/*
 * scalce_oracle.c -- plain-C CPU restatement of SCALCE 2.8's compress/decompress
 * hot path.  TEST INFRASTRUCTURE ONLY (see scalce_oracle.h for who may use it and
 * for the pinning status of each part).
 *
 * This is a restatement, not a copy: the reference keeps a pointer-linked trie,
 * per-bucket linked lists and temp files; this file keeps index arrays and
 * in-memory streams and reproduces the same observable results.  Each function
 * names the reference lines it follows (paths relative to /root/reference).
 */
#include "scalce_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------------- */
/* base -> 2 bit (const.cpp:47-49, const.h:127).  The reference indexes a     */
/* 58-entry table with c-'A'; bytes outside 'A'..'z' are undefined behaviour   */
/* there and are defined as 0 (same as 'A') here.                              */
/* ------------------------------------------------------------------------- */
static inline int base2(int c) {
  switch (c) {
  case 'C': case 'c': return 1;
  case 'G': case 'g': return 2;
  case 'T': case 't': return 3;
  default: return 0;
  }
}

#define SZ_READ(l) (((l) / 4) + ((l) % 4 > 0)) /* const.h:63 */

/* ------------------------------------------------------------------------- */
/* trie / automaton                                                           */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t child[4];  /* trie child, later DFA transition; -1 = none */
  int32_t fail;
  int32_t nto;       /* next_to_output; -1 = none */
  int32_t level;
  int32_t id;
  int32_t output;    /* pattern index or -1 */
  uint64_t bin_size; /* reads.h:82 */
} node_t;

struct orc_trie {
  node_t *nd;
  int n, cap;      /* n includes the root (index 0) */
  char **pat;
  int *plen;
  int *pnode;      /* node that ends pattern p */
  int np, pcap;
  int built;
};

static int trie_new_node(orc_trie *t, int level) {
  if (t->n == t->cap) {
    t->cap = t->cap ? t->cap * 2 : 1024;
    t->nd = (node_t *)realloc(t->nd, (size_t)t->cap * sizeof(node_t));
  }
  node_t *x = &t->nd[t->n];
  x->child[0] = x->child[1] = x->child[2] = x->child[3] = -1;
  x->fail = -1; x->nto = -1; x->level = level; x->id = 0; x->output = -1; x->bin_size = 0;
  return t->n++;
}

static orc_trie *trie_alloc(void) {
  orc_trie *t = (orc_trie *)calloc(1, sizeof(*t));
  trie_new_node(t, 0);
  return t;
}

/* pattern_insert, reads.cpp:253-267 (iterative; a later identical pattern
 * overwrites `output` exactly as the recursion's final assignment does) */
static void trie_insert(orc_trie *t, const char *s, int len) {
  if (t->np == t->pcap) {
    t->pcap = t->pcap ? t->pcap * 2 : 1024;
    t->pat = (char **)realloc(t->pat, (size_t)t->pcap * sizeof(char *));
    t->plen = (int *)realloc(t->plen, (size_t)t->pcap * sizeof(int));
    t->pnode = (int *)realloc(t->pnode, (size_t)t->pcap * sizeof(int));
  }
  int cur = 0;
  for (int i = 0; i < len; i++) {
    int c = base2((unsigned char)s[i]);
    if (t->nd[cur].child[c] < 0) {
      int nn = trie_new_node(t, t->nd[cur].level + 1);
      t->nd[cur].child[c] = nn;
    }
    cur = t->nd[cur].child[c];
  }
  t->nd[cur].output = t->np;
  t->pat[t->np] = (char *)malloc((size_t)len + 1);
  memcpy(t->pat[t->np], s, (size_t)len);
  t->pat[t->np][len] = 0;
  t->plen[t->np] = len;
  t->pnode[t->np] = cur;
  t->np++;
}

/* prepare_aho_automata, reads.cpp:270-315 */
static void trie_build(orc_trie *t) {
  int *q = (int *)malloc((size_t)(t->n + 1) * sizeof(int));
  int qh = 0, qt = 0;
  node_t *nd = t->nd;
  nd[0].fail = 0;
  for (int i = 0; i < 4; i++)
    if (nd[0].child[i] >= 0) {
      nd[nd[0].child[i]].fail = 0;
      q[qt++] = nd[0].child[i];
    }
  int traversed = 0;
  while (qh < qt) { /* pass 1: fail links, proper-suffix output link, BFS id (:282-297) */
    int cur = q[qh++];
    for (int i = 0; i < 4; i++) {
      int c = nd[cur].child[i];
      if (c < 0) continue;
      int f = nd[cur].fail;
      while (f != 0 && nd[f].child[i] < 0) f = nd[f].fail;
      nd[c].fail = nd[f].child[i] >= 0 ? nd[f].child[i] : 0;
      q[qt++] = c;
      int ff = nd[c].fail;
      nd[c].nto = nd[ff].output >= 0 ? ff : nd[ff].nto;
    }
    nd[cur].id = ++traversed;
  }
  qh = qt = 0;
  q[qt++] = 0;
  while (qh < qt) { /* pass 2: total transitions + self-or-nearest output (:300-315) */
    int cur = q[qh++];
    for (int i = 0; i < 4; i++) {
      if (nd[cur].child[i] >= 0) q[qt++] = nd[cur].child[i];
      int c = cur;
      while (c != 0 && nd[c].child[i] < 0) c = nd[c].fail;
      nd[cur].child[i] = nd[c].child[i] >= 0 ? nd[c].child[i] : 0;
      c = cur;
      while (c >= 0 && nd[c].output == -1) c = nd[c].nto;
      nd[cur].nto = c;
    }
  }
  free(q);
  t->built = 1;
}

/* read_patterns, reads.cpp:330-377: groups of [int16 ln][int32 cnt] then cnt
 * little-endian integers of ceil(ln/4) bytes, first base most significant */
orc_trie *orc_trie_from_bin(const uint8_t *blob, size_t n) {
  orc_trie *t = trie_alloc();
  size_t pos = 0;
  char buf[40];
  while (pos < n) {
    int16_t ln; int32_t cnt;
    if (pos + 6 > n) break;
    memcpy(&ln, blob + pos, 2); pos += 2;
    memcpy(&cnt, blob + pos, 4); pos += 4;
    int sz = ln / 4 + (ln % 4 != 0);
    if (ln <= 0 || ln > 32 || sz > 8) { orc_trie_free(t); return NULL; }
    for (int i = 0; i < cnt; i++) {
      if (pos + (size_t)sz > n) { orc_trie_free(t); return NULL; }
      uint64_t x = 0;
      memcpy(&x, blob + pos, (size_t)sz); pos += (size_t)sz;
      for (int j = 0; j < ln; j++) buf[j] = "ACGT"[(x >> (2 * (ln - 1 - j))) & 3];
      trie_insert(t, buf, ln);
    }
  }
  trie_build(t);
  return t;
}

/* read_patterns_from_file, reads.cpp:379-410: whitespace separated tokens */
orc_trie *orc_trie_from_text(const char *text, size_t n) {
  orc_trie *t = trie_alloc();
  size_t i = 0;
  while (i < n) {
    while (i < n && (text[i] == ' ' || text[i] == '\n' || text[i] == '\t' || text[i] == '\r')) i++;
    size_t s = i;
    while (i < n && !(text[i] == ' ' || text[i] == '\n' || text[i] == '\t' || text[i] == '\r')) i++;
    if (i > s) trie_insert(t, text + s, (int)(i - s));
  }
  trie_build(t);
  return t;
}

void orc_trie_free(orc_trie *t) {
  if (!t) return;
  for (int i = 0; i < t->np; i++) free(t->pat[i]);
  free(t->pat); free(t->plen); free(t->pnode); free(t->nd); free(t);
}
int orc_trie_patterns(const orc_trie *t) { return t->np; }
int orc_trie_nodes(const orc_trie *t) { return t->n - 1; }
int orc_trie_pattern_len(const orc_trie *t, int p) { return t->plen[p]; }
const char *orc_trie_pattern(const orc_trie *t, int p) { return t->pat[p]; }
int orc_trie_pattern_id(const orc_trie *t, int p) {
  int nd = t->pnode[p];
  return t->nd[nd].output == p ? t->nd[nd].id : -1;
}
void orc_trie_reset_counts(orc_trie *t) {
  for (int i = 0; i < t->n; i++) t->nd[i].bin_size = 0;
}

/* aho_search, reads.cpp:413-429 */
static int search_node(const orc_trie *t, const char *text, int L, int *node_out) {
  const node_t *nd = t->nd;
  int cur = 0, largest = -1, bestpos = -1;
  for (int i = 0; i < L; i++) {
    cur = nd[cur].child[base2((unsigned char)text[i])];
    int x = nd[cur].nto;
    if (x >= 0) {
      if (largest < 0 || nd[largest].level < nd[x].level ||
          (nd[largest].level == nd[x].level && nd[largest].bin_size < nd[x].bin_size)) {
        bestpos = i;
        largest = x;
      }
    }
  }
  *node_out = largest < 0 ? 0 : largest;
  return bestpos;
}

int orc_search(const orc_trie *t, const char *text, int L, int *pattern, int *level) {
  int nodei;
  int bp = search_node(t, text, L, &nodei);
  *pattern = t->nd[nodei].output;
  *level = t->nd[nodei].level;
  return bp;
}

void orc_bucket_bump(orc_trie *t, int pattern) {
  int nodei = pattern < 0 ? 0 : t->pnode[pattern];
  t->nd[nodei].bin_size++;
}

/* output_read, reads.cpp:432-461 */
int orc_pack_read(const char *line, int L, int n, int l, uint8_t *dest) {
  int bc = 0, cc = 0;
  uint8_t ca = 0;
  for (int i = n + l; i < L; i++) {
    ca = (uint8_t)((ca << 2) | base2((unsigned char)line[i]));
    if (++cc == 4) { dest[bc++] = ca; cc = 0; }
  }
  for (int i = 0; i < n; i++) {
    ca = (uint8_t)((ca << 2) | base2((unsigned char)line[i]));
    if (++cc == 4) { dest[bc++] = ca; cc = 0; }
  }
  if (cc) {
    while (cc != 4) { ca = (uint8_t)(ca << 2); cc++; }
    dest[bc++] = ca;
  }
  return bc;
}

/* output_name, names.cpp:48-62 with _interleave == 0 */
int orc_pack_name(const char *name, int use_names, uint8_t *dest) {
  if (!use_names) { dest[0] = 0; return 1; }
  int i;
  for (i = 1; name[i] != '\n' && name[i] != ' ' && name[i] != 0; i++) dest[i] = (uint8_t)name[i];
  dest[0] = (uint8_t)(i - 1);
  return dest[0] + 1;
}

/* the per-read part of thread(), compress.cpp:673-701, at -T 1 */
void orc_tokenize_seq(orc_trie *t, const char *reads, int64_t N, int L, int stride,
                      int32_t *pattern_out, int32_t *end_out) {
  for (int64_t r = 0; r < N; r++) {
    int nodei;
    int bp = search_node(t, reads + r * (int64_t)stride, L, &nodei);
    pattern_out[r] = t->nd[nodei].output;
    end_out[r] = bp == -1 ? 0 : bp + 1; /* compress.cpp:682,685 */
    t->nd[nodei].bin_size++;            /* reads.cpp:246 */
  }
}

/* ---- in-bucket order: _radix_sort / bin_prepare, reads.cpp:547-634 ---- */
typedef struct {
  const char *reads;
  int L, stride;
  const int32_t *end;
  int limit;
  int64_t *nodes, *temp;
} rsort_t;

/* _POS, reads.cpp:557-558: stored base `pos` of the rotated read while it lies in
 * the suffix-after-core part, else 0.  Stored base pos == original base end+pos. */
static inline int rs_digit(const rsort_t *s, int64_t rd, int pos) {
  int e = s->end[rd];
  return (pos + e < s->L) ? base2((unsigned char)s->reads[rd * (int64_t)s->stride + e + pos]) : 0;
}

static void rs_sort(rsort_t *s, int pos, int64_t start, int64_t size) {
  if (size <= 1 || pos >= s->limit) return;
  int64_t count[4] = {0, 0, 0, 0}, cum[5];
  for (int64_t i = start; i < start + size; i++) {
    count[rs_digit(s, s->nodes[i], pos)]++;
    s->temp[i] = s->nodes[i];
  }
  cum[0] = 0;
  for (int i = 1; i < 5; i++) cum[i] = cum[i - 1] + count[i - 1];
  int64_t w[4] = {cum[0], cum[1], cum[2], cum[3]};
  for (int64_t i = start; i < start + size; i++) {
    int c = rs_digit(s, s->temp[i], pos);
    s->nodes[start + w[c]++] = s->temp[i];
  }
  for (int i = 0; i < 4; i++) rs_sort(s, pos + 1, start + cum[i], cum[i + 1] - cum[i]);
}

void orc_bucket_order(const orc_trie *t, const char *reads, int64_t N, int L, int stride,
                      const int32_t *pattern, const int32_t *end, const int32_t *chunk,
                      int64_t *perm_out) {
  /* bucket emission order = BFS over the automaton = increasing id, root last
   * (aho_output, reads.cpp:466-499); chunks concatenate per bucket in chunk
   * order (merhamet_merge, compress.cpp:104-159). */
  int nid = t->n + 1;
  int64_t *cnt = (int64_t *)calloc((size_t)nid + 1, sizeof(int64_t));
  int32_t *key = (int32_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int32_t));
  for (int64_t r = 0; r < N; r++) {
    int k = pattern[r] < 0 ? nid - 1 : t->nd[t->pnode[pattern[r]]].id - 1;
    key[r] = k;
    cnt[k + 1]++;
  }
  for (int i = 0; i < nid; i++) cnt[i + 1] += cnt[i];
  int64_t *pos = (int64_t *)malloc((size_t)nid * sizeof(int64_t));
  memcpy(pos, cnt, (size_t)nid * sizeof(int64_t));
  for (int64_t r = 0; r < N; r++) perm_out[pos[key[r]]++] = r; /* stable: input order */
  rsort_t s;
  s.reads = reads; s.L = L; s.stride = stride; s.end = end;
  s.nodes = perm_out;
  s.temp = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
  for (int b = 0; b < nid; b++) {
    int64_t lo = cnt[b], hi = cnt[b + 1];
    if (hi <= lo) continue;
    int level = 0;
    if (b != nid - 1) {
      int p = pattern[perm_out[lo]];
      level = t->plen[p];
    }
    s.limit = L - level; /* reads.cpp:625 */
    int64_t i = lo;
    while (i < hi) { /* one sort per spill chunk (dump_trie per chunk, compress.cpp:708-715) */
      int64_t j = i;
      int c = chunk ? chunk[perm_out[i]] : 0;
      while (j < hi && (chunk ? chunk[perm_out[j]] : 0) == c) j++;
      rs_sort(&s, 0, i, j - i);
      i = j;
    }
  }
  free(s.temp); free(pos); free(key); free(cnt);
}

/* ------------------------------------------------------------------------- */
/* qualities                                                                  */
/* ------------------------------------------------------------------------- */
static double phred(int c, int offset) { return pow(10, -(c - offset) / 10.0); } /* qualities.cpp:53 */

/* quality_mapping_init after sampling, qualities.cpp:99-174 */
void orc_qmap_init(orc_qmap *q, const int stat[128], int lossy) {
  q->offset = 64;
  for (int i = 33; i < 64; i++)
    if (stat[i]) { q->offset = 33; break; }
  for (int c = 0; c < 128; c++) q->values[c] = c;
  if (!lossy) return;

  char assigned[128];
  memset(assigned, 0, sizeof assigned);
  for (int h = q->offset; 100 * phred(h, q->offset) > 30; h++) { /* :119-122 */
    q->values[h] = q->offset;
    assigned[h] = 1;
  }
  /* order symbols by (count desc, char asc), :125-140 */
  int ord[128];
  for (int i = 0; i < 128; i++) ord[i] = i;
  for (int i = 0; i < 128; i++)
    for (int j = i + 1; j < 128; j++)
      if (stat[ord[j]] > stat[ord[i]] || (stat[ord[j]] == stat[ord[i]] && ord[j] < ord[i])) {
        int x = ord[i]; ord[i] = ord[j]; ord[j] = x;
      }
  double pct = lossy / 100.0;
  for (int i = 0; i < 128 && stat[ord[i]]; i++) { /* :144-174 */
    int c = ord[i];
    int sl = c > 0 ? stat[c - 1] : 0, sr = c < 127 ? stat[c + 1] : 0;
    if (assigned[c] || stat[c] < sl || stat[c] < sr) continue;
    double er = phred(c, q->offset), total = er;
    int left = c, right = c;
    for (int k = c - 1; k >= 0; k--) {
      total += phred(k, q->offset);
      if (assigned[k] || total / (c - k + 1) > er + (er * pct)) { left = k + 1; break; }
    }
    total = er;
    for (int k = c + 1; k < 128; k++) {
      total += phred(k, q->offset);
      if (assigned[k] || total / (k - c + 1) < er - (er * pct)) { right = k - 1; break; }
    }
    for (int k = left; k <= right; k++) { q->values[k] = c; assigned[k] = 1; }
  }
}

/* output_quality, qualities.cpp:177-204 */
void orc_quality(const char *qual, const char *read, int L, const orc_qmap *q, uint8_t *dest,
                 uint64_t *freq4, uint32_t state[2], int no_ac) {
  for (int l = 0; l < L; l++) {
    int v = (read[l] == 'N' ? q->offset : q->values[(unsigned char)qual[l] & 127]) - q->offset;
    dest[l] = (uint8_t)v;
    if (!no_ac) {
      if (state[1] < 256) {
        if (state[0] < 256)
          freq4[((uint64_t)state[0] * ORC_AC_DEPTH + state[1]) * ORC_AC_DEPTH + dest[l]]++;
      } else { /* very first symbol of this mate: every counter := 1 (:191-196) */
        for (int e = 0; e < ORC_AC_DEPTH * ORC_AC_DEPTH * ORC_AC_DEPTH; e++) freq4[e] = 1;
      }
      state[0] = state[1];
      state[1] = dest[l];
    }
  }
}

/* ------------------------------------------------------------------------- */
/* arithmetic coder                                                           */
/* ------------------------------------------------------------------------- */
void orc_ac_scale(const uint64_t *freq4, int factor, uint32_t *out) { /* compress.cpp:306-320 */
  for (int i = 0; i < ORC_AC_DEPTH * ORC_AC_DEPTH * ORC_AC_DEPTH; i++) {
    uint64_t p = freq4[i] / (uint64_t)factor;
    if (p == 0) p = 1;
    out[i] = (uint32_t)p;
  }
}

void orc_acstat_init(orc_acstat *s, const uint32_t *table) { /* arithmetic.cpp:54-78 */
  s->cnt = table;
  for (int c = 0; c < ORC_AC_DEPTH * ORC_AC_DEPTH; c++) {
    const uint32_t *f = table + (size_t)c * ORC_AC_DEPTH;
    s->hi[c][0] = f[0];
    for (int l = 1; l < ORC_AC_DEPTH; l++) s->hi[c][l] = s->hi[c][l - 1] + f[l];
    s->tot[c] = s->hi[c][ORC_AC_DEPTH - 1];
    int p = -1;
    for (int l = 0; l < ORC_AC_DEPTH; l++) s->lo[c][l] = 0;
    for (int l = 0; l < ORC_AC_DEPTH; l++)
      if (f[l]) {
        if (p != -1) s->lo[c][l] = s->hi[c][p];
        p = l;
      }
  }
}

typedef struct {
  uint8_t *ou;
  int buf_pos;
} bitw_t;

static inline void put_bit(bitw_t *w, int b) { /* ac_coder::O, arithmetic.cpp:94-106 */
  *w->ou = (uint8_t)((*w->ou << 1) | (b ? 1 : 0));
  if (w->buf_pos == 0) { w->ou++; w->buf_pos = 7; *w->ou = 0; }
  else w->buf_pos--;
}

/* ac_coder::reset + write + flush, arithmetic.cpp:85-169 */
size_t orc_ac_encode_block(const orc_acstat *s, const uint8_t *a, size_t n, uint8_t *out) {
  uint32_t lo = 0, hi = 0xFFFFFFFFu, underflow = 0;
  bitw_t w; w.ou = out; w.buf_pos = 7;
  uint32_t p0, p1;
  size_t i = 0;
  /* the first two symbols of a block travel raw (:110-120).  A 1-symbol block makes
   * the reference read one byte past its input; that byte is defined as 0 here. */
  out[0] = a[0]; out[1] = n > 1 ? a[1] : 0;
  p0 = out[0]; p1 = out[1];
  w.ou = out + 2;
  *w.ou = 0;
  i = 2;
  for (; i < n; i++) {
    uint32_t ctx = p0 * ORC_AC_DEPTH + p1;
    uint32_t c_lo = s->lo[ctx][a[i]], c_hi = s->hi[ctx][a[i]], p_sz = s->tot[ctx];
    uint64_t range = (uint64_t)(uint32_t)(hi - lo) + 1;
    hi = (uint32_t)(lo + (range * c_hi) / p_sz - 1);
    lo = (uint32_t)(lo + (range * c_lo) / p_sz);
    for (;;) {
      if ((hi & 0x80000000u) == (lo & 0x80000000u)) {
        put_bit(&w, hi & 0x80000000u);
        while (underflow) { put_bit(&w, ~hi & 0x80000000u); underflow--; }
      } else if (!(hi & 0x40000000u) && (lo & 0x40000000u)) {
        underflow++;
        lo &= 0x3FFFFFFFu;
        hi |= 0x40000000u;
      } else break;
      lo <<= 1;
      hi = (hi << 1) | 1;
    }
    p0 = p1; p1 = a[i];
  }
  put_bit(&w, lo & 0x40000000u); /* flush, :160-169 */
  underflow++;
  while (underflow) { put_bit(&w, ~lo & 0x40000000u); underflow--; }
  while (w.buf_pos != 7) put_bit(&w, 0);
  return (size_t)(w.ou - out);
}

/* ac_decoder::read / read_single, arithmetic.cpp:177-268 */
void orc_ac_decode_block(const orc_acstat *s, const uint8_t *in, size_t nsym, uint8_t *ar) {
  if (nsym == 0) return;
  uint32_t lo = 0, hi = 0xFFFFFFFFu, code = 0;
  uint32_t p0 = in[0], p1 = in[1];
  ar[0] = in[0];
  if (nsym > 1) ar[1] = in[1];
  const uint8_t *ip = in + 2;
  int bp = 7;
#define GETBIT() (bp == 0 ? (bp = 7, (*ip++) & 1) : (((*ip) >> bp--) & 1))
  for (int j = 0; j < 32; j++) { int b = GETBIT(); code = (code << 1) | (uint32_t)b; }
  for (size_t i = 2; i < nsym; i++) {
    uint32_t ctx = p0 * ORC_AC_DEPTH + p1;
    uint64_t range = (uint64_t)(uint32_t)(hi - lo) + 1;
    uint32_t count = (uint32_t)((((uint64_t)(uint32_t)(code - lo) + 1) * (uint64_t)s->tot[ctx] - 1) / range);
    uint32_t k;
    const uint32_t *f = s->cnt + (size_t)ctx * ORC_AC_DEPTH;
    for (k = 0; k < ORC_AC_DEPTH; k++)
      if (f[k] && count >= s->lo[ctx][k] && count < s->hi[ctx][k]) break;
    if (k == ORC_AC_DEPTH) k = ORC_AC_DEPTH - 1; /* reference asserts; corrupt stream */
    hi = (uint32_t)(lo + (range * s->hi[ctx][k]) / s->tot[ctx] - 1);
    lo = (uint32_t)(lo + (range * s->lo[ctx][k]) / s->tot[ctx]);
    for (;;) {
      if ((hi & 0x80000000u) == (lo & 0x80000000u)) {
      } else if (!(hi & 0x40000000u) && (lo & 0x40000000u)) {
        code ^= 0x40000000u;
        lo &= 0x3FFFFFFFu;
        hi |= 0x40000000u;
      } else break;
      lo <<= 1;
      hi = (hi << 1) | 1;
      { int b = GETBIT(); code = (code << 1) | (uint32_t)b; }
    }
    p0 = p1; p1 = k;
    ar[i] = (uint8_t)k;
  }
#undef GETBIT
}

/* ac_write + thread_c, arithmetic.cpp:280-287,318-363: independent 10 MiB blocks,
 * each emitted as [u32 nbytes][bytes]; boundaries do not depend on the thread count. */
typedef struct {
  const orc_acstat *s;
  const uint8_t *sym;
  size_t n, nblk, cap;
  uint8_t *scratch;
  uint32_t *sizes;
  size_t next;
  pthread_mutex_t mu;
} acjob_t;

static void *ac_worker(void *v) {
  acjob_t *j = (acjob_t *)v;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    size_t b = j->next++;
    pthread_mutex_unlock(&j->mu);
    if (b >= j->nblk) return 0;
    size_t off = b * (size_t)ORC_AC_BLOCK;
    size_t len = j->n - off < (size_t)ORC_AC_BLOCK ? j->n - off : (size_t)ORC_AC_BLOCK;
    j->sizes[b] = (uint32_t)orc_ac_encode_block(j->s, j->sym + off, len, j->scratch + b * j->cap);
  }
}

size_t orc_ac_encode_stream(const orc_acstat *s, const uint8_t *sym, size_t n, uint8_t *out,
                            size_t out_cap, int threads) {
  if (n == 0) return 0;
  acjob_t j;
  j.s = s; j.sym = sym; j.n = n;
  j.nblk = (n + ORC_AC_BLOCK - 1) / ORC_AC_BLOCK;
  j.cap = (size_t)ORC_AC_BLOCK * 2 + 64;
  j.scratch = (uint8_t *)malloc(j.nblk * j.cap);
  j.sizes = (uint32_t *)malloc(j.nblk * sizeof(uint32_t));
  j.next = 0;
  pthread_mutex_init(&j.mu, 0);
  if (threads < 1) threads = 1;
  if ((size_t)threads > j.nblk) threads = (int)j.nblk;
  pthread_t th[64];
  if (threads > 64) threads = 64;
  for (int i = 1; i < threads; i++) pthread_create(&th[i], 0, ac_worker, &j);
  ac_worker(&j);
  for (int i = 1; i < threads; i++) pthread_join(th[i], 0);
  size_t pos = 0;
  for (size_t b = 0; b < j.nblk; b++) {
    if (pos + 4 + j.sizes[b] > out_cap) { pos = (size_t)-1; break; }
    memcpy(out + pos, &j.sizes[b], 4); pos += 4;
    memcpy(out + pos, j.scratch + b * j.cap, j.sizes[b]); pos += j.sizes[b];
  }
  free(j.scratch); free(j.sizes);
  pthread_mutex_destroy(&j.mu);
  return pos;
}

/* ------------------------------------------------------------------------- */
/* file pipelines                                                             */
/* ------------------------------------------------------------------------- */
void orc_opts_default(orc_opts *o) {
  memset(o, 0, sizeof *o);
  o->use_names = 1;
  o->library = "";
  o->sample = 100000;                   /* main.cpp:62 */
  o->gz = 1;                            /* main.cpp:184 at -T 1 */
  o->bucket_set_size = 4ull << 30;      /* main.cpp:68 */
  o->threads = 1;
}

#define MAXLINE 2500 /* const.h:87 */

/* get_second_file, const.cpp:51-64: last '1' -> '2' */
static int second_file(const char *c, char *buf, size_t n) {
  snprintf(buf, n, "%s", c);
  for (int i = (int)strlen(buf) - 1; i >= 0; i--)
    if (buf[i] == '1') { buf[i] = '2'; return 1; }
  return 0;
}

/* sampling half of quality_mapping_init, qualities.cpp:64-97 */
static int sample_stats(const char *path, int sample, int stat[128], int *read_length) {
  gzFile f = gzopen(path, "rb");
  if (!f) { fprintf(stderr, "(ERROR) Cannot open file %s!\n", path); return -1; }
  char line[MAXLINE];
  memset(stat, 0, 128 * sizeof(int));
  for (int i = 0; i < sample; i++) {
    gzgets(f, line, MAXLINE); gzgets(f, line, MAXLINE); gzgets(f, line, MAXLINE);
    if (!gzgets(f, line, MAXLINE)) break;
    int l = (int)strlen(line) - 1;
    for (int j = 0; j < l; j++) stat[(unsigned char)line[j] & 127]++;
    *read_length = l;
  }
  gzclose(f);
  return 0;
}

typedef struct {
  uint8_t *p;
  size_t n, cap;
} buf_t;
static void buf_put(buf_t *b, const void *src, size_t n) {
  if (b->n + n > b->cap) {
    b->cap = (b->n + n) * 2 + 4096;
    b->p = (uint8_t *)realloc(b->p, b->cap);
  }
  memcpy(b->p + b->n, src, n);
  b->n += n;
}

typedef struct {
  int gz;
  gzFile g;
  FILE *f;
} ofile_t;
static int of_open(ofile_t *o, const char *path, int gz) {
  o->gz = gz; o->g = 0; o->f = 0;
  if (gz) o->g = gzopen(path, "wb"); else o->f = fopen(path, "wb");
  return (o->g || o->f) ? 0 : -1;
}
static void of_write(ofile_t *o, const void *p, size_t n) {
  while (n) {
    size_t k = n > (1u << 30) ? (1u << 30) : n;
    if (o->gz) gzwrite(o->g, p, (unsigned)k); else fwrite(p, 1, k, o->f);
    p = (const uint8_t *)p + k; n -= k;
  }
}
static void of_close(ofile_t *o) { if (o->gz) gzclose(o->g); else fclose(o->f); }

int orc_compress_files(orc_trie *t, const char *fastq1, const char *out_prefix, const orc_opts *o) {
  char path2[4096], line[MAXLINE], name[MAXLINE];
  orc_qmap qm[2];
  int rl[2] = {0, 0}, stat[128];
  const int nm = o->paired ? 2 : 1;
  if (o->paired && !second_file(fastq1, path2, sizeof path2)) {
    fprintf(stderr, "(ERROR) Cannot get file name for paired end for file %s.\n", fastq1);
    return 1;
  }
  /* get_quality_stats, compress.cpp:554-582 */
  if (sample_stats(fastq1, o->sample, stat, &rl[0])) return 1;
  orc_qmap_init(&qm[0], stat, o->lossy);
  if (o->paired) {
    if (sample_stats(path2, o->sample, stat, &rl[1])) return 1;
    orc_qmap_init(&qm[1], stat, o->lossy);
  }
  gzFile in[2] = {0, 0};
  in[0] = gzopen(fastq1, "rb");
  if (o->paired) in[1] = gzopen(path2, "rb");
  if (!in[0] || (o->paired && !in[1])) { fprintf(stderr, "(ERROR) Cannot read file %s\n", fastq1); return 1; }
  gzbuffer(in[0], 1 << 20);
  if (in[1]) gzbuffer(in[1], 1 << 20);

  orc_trie_reset_counts(t);
  /* per-read records, kept in memory in input order (the reference keeps them in
   * per-bucket lists, reads.cpp:233-250; the order is rebuilt by orc_bucket_order) */
  buf_t bases[2] = {{0, 0, 0}, {0, 0, 0}}, quals[2] = {{0, 0, 0}, {0, 0, 0}}, names = {0, 0, 0};
  buf_t nameoff = {0, 0, 0}, pat = {0, 0, 0}, endv = {0, 0, 0}, chunkv = {0, 0, 0};
  uint64_t *freq4[2] = {0, 0};
  uint32_t qstate[2][2] = {{500, 500}, {500, 500}};
  for (int m = 0; m < nm; m++) freq4[m] = (uint64_t *)calloc(512000, sizeof(uint64_t));
  int64_t N = 0;
  uint64_t total_size = 0;
  int32_t chunk = 0;
  int xlen[2] = {0, 0};
  uint8_t nb[MAXLINE], qb[MAXLINE];
  char rd[2][MAXLINE], ql[2][MAXLINE];
  while (gzgets(in[0], name, MAXLINE)) { /* thread(), compress.cpp:613-716 */
    if (!gzgets(in[0], rd[0], MAXLINE)) break;
    int l = (int)strlen(rd[0]);
    if (!l || rd[0][0] == '\n') { fprintf(stderr, "(ERROR) empty read\n"); return 1; }
    if (!xlen[0]) xlen[0] = l;
    else if (l != xlen[0]) { fprintf(stderr, "Whooops... read names in /1 do not match (%d vs %d)!\n", xlen[0] - 1, l - 1); return 1; }
    gzgets(in[0], ql[0], MAXLINE); gzgets(in[0], ql[0], MAXLINE);
    if (o->paired) {
      gzgets(in[1], rd[1], MAXLINE); gzgets(in[1], rd[1], MAXLINE);
      int l2 = (int)strlen(rd[1]);
      if (!xlen[1]) xlen[1] = l2;
      else if (l2 != xlen[1]) { fprintf(stderr, "Whooops... read names in /2 do not match (%d vs %d)!\n", xlen[1] - 1, l2 - 1); return 1; }
      gzgets(in[1], ql[1], MAXLINE); gzgets(in[1], ql[1], MAXLINE);
    }
    /* the reference walks to '\n' (reads.cpp:416, qualities.cpp:182); with a well
     * formed file that is rl[m] characters */
    int nodei, level, p;
    int bp = search_node(t, rd[0], rl[0], &nodei);
    p = t->nd[nodei].output; level = t->nd[nodei].level;
    int32_t e = bp == -1 ? 0 : bp + 1;
    t->nd[nodei].bin_size++;
    int nsz = orc_pack_name(name, o->use_names, nb);
    uint64_t off = names.n;
    buf_put(&nameoff, &off, 8);
    buf_put(&names, nb, (size_t)nsz);
    uint64_t sz = (uint64_t)nsz + SZ_READ(rl[0] - (bp == -1 ? 0 : level));
    for (int m = 0; m < nm; m++) {
      orc_quality(ql[m], rd[m], rl[m], &qm[m], qb, freq4[m], qstate[m], o->no_ac);
      buf_put(&quals[m], qb, (size_t)rl[m]);
      buf_put(&bases[m], rd[m], (size_t)rl[m]);
      sz += (uint64_t)rl[m];
      if (m) sz += SZ_READ(rl[1]);
    }
    buf_put(&pat, &p, 4); buf_put(&endv, &e, 4); buf_put(&chunkv, &chunk, 4);
    N++;
    total_size += sz + 40; /* sizeof(bin_node), compress.cpp:702 */
    if (total_size >= o->bucket_set_size) { chunk++; total_size = 0; } /* :708-715 */
  }
  gzclose(in[0]);
  if (in[1]) gzclose(in[1]);
  { uint64_t off = names.n; buf_put(&nameoff, &off, 8); }

  int64_t *perm = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
  const int32_t *patv = (const int32_t *)pat.p, *ev = (const int32_t *)endv.p;
  orc_bucket_order(t, (const char *)bases[0].p, N, rl[0], rl[0], patv, ev, (const int32_t *)chunkv.p, perm);

  /* combine_and_compress_with_split, compress.cpp:200-486 */
  const uint8_t magic[8] = {'s', 'c', 'a', 'l', 'c', 'e', '2', '2'};
  const int sz_meta = rl[0] > 255 ? 2 : 1;
  for (int m = 0; m < nm; m++) {
    char fn[4096];
    ofile_t fR, fN, fQ;
    snprintf(fn, sizeof fn, "%s_%d.scalcen", out_prefix, m + 1); if (of_open(&fN, fn, o->gz)) return 1;
    snprintf(fn, sizeof fn, "%s_%d.scalceq", out_prefix, m + 1); if (of_open(&fQ, fn, o->no_ac ? o->gz : 0)) return 1;
    snprintf(fn, sizeof fn, "%s_%d.scalcer", out_prefix, m + 1); if (of_open(&fR, fn, o->gz)) return 1;
    int32_t noac = o->no_ac, len32 = rl[m];
    of_write(&fR, magic, 8); of_write(&fR, &noac, 4); of_write(&fR, &len32, 4);
    int64_t phred_off = qm[0].offset; /* mate 1's offset for both, compress.cpp:294,816-817 */
    of_write(&fQ, magic, 8); of_write(&fQ, &phred_off, 8);
    uint32_t *table = 0;
    orc_acstat *as = 0;
    if (!o->no_ac) {
      int factor = 1 + (int)(((uint64_t)N * (uint64_t)rl[m]) / 0xFFFFFFFFull); /* :297-303 */
      table = (uint32_t *)malloc(512000 * sizeof(uint32_t));
      orc_ac_scale(freq4[m], factor, table);
      of_write(&fQ, table, 512000 * sizeof(uint32_t));
      uint64_t tot = (uint64_t)N * (uint64_t)rl[m];
      of_write(&fQ, &tot, 8);
      as = (orc_acstat *)malloc(sizeof(orc_acstat));
      orc_acstat_init(as, table);
    }
    uint8_t use_names = (uint8_t)o->use_names;
    of_write(&fN, magic, 8); of_write(&fN, &use_names, 1);
    if (!o->use_names) {
      int64_t z = 0;
      of_write(&fN, &z, 8); of_write(&fN, o->library, strlen(o->library));
    }
    /* reordered quality stream */
    uint8_t *qs = (uint8_t *)malloc((size_t)N * (size_t)rl[m] + 1);
    for (int64_t k = 0; k < N; k++)
      memcpy(qs + (size_t)k * (size_t)rl[m], quals[m].p + (size_t)perm[k] * (size_t)rl[m], (size_t)rl[m]);
    /* reads + names, bucket by bucket */
    uint8_t pk[MAXLINE];
    int64_t k = 0;
    while (k < N) {
      int64_t j = k;
      int32_t p = patv[perm[k]];
      while (j < N && patv[perm[j]] == p) j++;
      int level = p < 0 ? 0 : t->plen[p];
      if (m == 0) {
        int32_t core = p < 0 ? ORC_ROOT_CORE : p;
        int64_t cnt = j - k;
        of_write(&fR, &core, 4); of_write(&fR, &cnt, 8); /* :364-379 */
      }
      for (int64_t i = k; i < j; i++) {
        int64_t r = perm[i];
        const char *b = (const char *)bases[m].p + (size_t)r * (size_t)rl[m];
        if (m == 0) {
          int32_t e = ev[r];
          int n = e ? e - level : 0, l = e ? level : 0;
          int nb2 = orc_pack_read(b, rl[0], n, l, pk);
          of_write(&fR, pk, (size_t)nb2);
          of_write(&fR, &e, (size_t)sz_meta); /* reads.cpp:130 */
        } else {
          int nb2 = orc_pack_read(b, rl[1], 0, 0, pk);
          of_write(&fR, pk, (size_t)nb2);
        }
        if (o->use_names) { /* mate 2's file repeats mate 1's names, compress.cpp:450-454 */
          uint64_t a = ((uint64_t *)nameoff.p)[r], z = ((uint64_t *)nameoff.p)[r + 1];
          of_write(&fN, names.p + a, (size_t)(z - a));
        }
      }
      k = j;
    }
    if (!o->no_ac) {
      size_t cap = (size_t)N * (size_t)rl[m] * 2 + 4096;
      uint8_t *enc = (uint8_t *)malloc(cap);
      size_t nb2 = orc_ac_encode_stream(as, qs, (size_t)N * (size_t)rl[m], enc, cap, o->threads);
      of_write(&fQ, enc, nb2);
      free(enc);
    } else {
      of_write(&fQ, qs, (size_t)N * (size_t)rl[m]);
    }
    free(qs); free(table); free(as);
    of_close(&fR); of_close(&fN); of_close(&fQ);
  }
  if (o->verbose) fprintf(stderr, "oracle: %lld reads, %d chunk(s)\n", (long long)N, chunk + (total_size ? 1 : 0));
  for (int m = 0; m < 2; m++) { free(bases[m].p); free(quals[m].p); free(freq4[m]); }
  free(names.p); free(nameoff.p); free(pat.p); free(endv.p); free(chunkv.p); free(perm);
  (void)line;
  return 0;
}

/* ---- decompress, decompress.cpp:79-397 ---- */
typedef struct {
  gzFile g; /* zlib reads plain files transparently, like the reference's sniffing (:99-113) */
} ifile_t;
static int64_t if_read(ifile_t *f, void *p, int64_t n) {
  int64_t got = 0;
  while (got < n) {
    int k = gzread(f->g, (uint8_t *)p + got, (unsigned)((n - got) > (1 << 30) ? (1 << 30) : (n - got)));
    if (k <= 0) break;
    got += k;
  }
  return got;
}

static void scalce_name(char *dst, size_t n, const char *path, char c) { /* get_file_name, :72-77 */
  snprintf(dst, n, "%s", path);
  char *p = 0, *h = dst;
  for (;;) { char *x = strstr(h, ".scalce"); if (!x) break; p = x; h = x + 1; }
  if (p) p[7] = c;
}

int orc_decompress_files(const orc_trie *t, const char *path, const char *out_prefix, const orc_opts *o) {
  const int nm = o->paired ? 2 : 1;
  char base[2][4096], fn[4096];
  snprintf(base[0], sizeof base[0], "%s", path);
  if (o->paired && !second_file(path, base[1], sizeof base[1])) return 1;
  ifile_t fR[2], fQ[2], fN[2];
  int32_t len[2] = {0, 0}, no_ac = 0;
  int64_t phred[2] = {0, 0};
  uint32_t *table[2] = {0, 0};
  uint64_t qtotal[2] = {0, 0};
  uint8_t b8[16];
  for (int m = 0; m < nm; m++) {
    scalce_name(fn, sizeof fn, base[m], 'r'); fR[m].g = gzopen(fn, "rb");
    if (!fR[m].g) { fprintf(stderr, "(ERROR) Cannot find read file %s!\n", fn); return 1; }
    scalce_name(fn, sizeof fn, base[m], 'n'); fN[m].g = gzopen(fn, "rb");
    scalce_name(fn, sizeof fn, base[m], 'q'); fQ[m].g = gzopen(fn, "rb");
    if (!fN[m].g || !fQ[m].g) { fprintf(stderr, "(ERROR) Cannot find name/quality file for %s!\n", fn); return 1; }
    if_read(&fR[m], b8, 8);
    no_ac = 0;
    if (b8[6] == '2' && b8[7] >= '2') if_read(&fR[m], &no_ac, 4); /* :149-151 */
    if_read(&fQ[m], b8, 8);
    if_read(&fN[m], b8, 8);
    if_read(&fR[m], &len[m], 4);
    if_read(&fQ[m], &phred[m], 8);
    if (!no_ac) {
      table[m] = (uint32_t *)malloc(512000 * 4);
      if_read(&fQ[m], table[m], 512000 * 4);
    }
  }
  uint8_t names = 0;
  char library[MAXLINE + 1];
  snprintf(library, sizeof library, "%s", o->library ? o->library : "");
  if (o->use_names) { /* :219-237 */
    for (int m = 0; m < nm; m++) if_read(&fN[m], &names, 1);
    if (!names)
      for (int m = 0; m < nm; m++) {
        int64_t idx;
        if_read(&fN[m], &idx, 8);
        int64_t k = if_read(&fN[m], library, MAXLINE);
        library[k] = 0;
      }
  }
  const int sz_meta = len[0] > 255 ? 2 : 1;
  int64_t next_info = 0;
  int32_t core = 0, corlen = 0; /* NOT reset between mates: the reference's leak (:250) */
  for (int m = 0; m < nm; m++) {
    snprintf(fn, sizeof fn, "%s_%d.fastq", out_prefix, m + 1);
    FILE *fo = fopen(fn, "wb");
    if (!fo) return 1;
    uint8_t *qs = 0;
    int64_t qpos = 0;
    if (!no_ac) { /* ac_read, arithmetic.cpp:365-400: all blocks decoded up front here */
      if_read(&fQ[m], &qtotal[m], 8);
      orc_acstat *as = (orc_acstat *)malloc(sizeof *as);
      orc_acstat_init(as, table[m]);
      qs = (uint8_t *)malloc(qtotal[m] + 1);
      uint8_t *blk = (uint8_t *)malloc((size_t)ORC_AC_BLOCK * 2 + 64);
      uint64_t left = qtotal[m], done = 0;
      while (left) {
        uint32_t bsz;
        if (if_read(&fQ[m], &bsz, 4) != 4) break;
        if_read(&fQ[m], blk, bsz);
        blk[bsz] = blk[bsz + 1] = blk[bsz + 2] = blk[bsz + 3] = 0;
        uint64_t ns = left < (uint64_t)ORC_AC_BLOCK ? left : (uint64_t)ORC_AC_BLOCK;
        orc_ac_decode_block(as, blk, ns, qs + done);
        done += ns; left -= ns;
      }
      free(blk); free(as);
    }
    uint8_t ob[MAXLINE], l[MAXLINE], qb[MAXLINE], nbuf[MAXLINE];
    int64_t nameidx = 0;
    for (int64_t K = 0;; K++) {
      if (K == next_info && m == 0) { /* :262-272 */
        uint64_t cnt = 0;
        if (if_read(&fR[0], &core, 4) != 4) break;
        if_read(&fR[0], &cnt, 8);
        next_info += (int64_t)cnt;
        corlen = core == ORC_ROOT_CORE ? 0 : t->plen[core];
      } else if (m && K == next_info) break;
      int n;
      if (names) { /* :290-299 */
        uint8_t chr;
        if_read(&fN[m], &chr, 1);
        nbuf[0] = '@';
        if_read(&fN[m], nbuf + 1, chr);
        if (o->paired && chr > 0 && nbuf[chr - 1] == '/') nbuf[chr] = (uint8_t)(m + 1 + '0');
        nbuf[chr + 1] = '\n';
        fwrite(nbuf, 1, (size_t)chr + 2, fo);
      } else {
        n = snprintf((char *)nbuf, MAXLINE, "@%s.%lld\n", library, (long long)nameidx);
        fwrite(nbuf, 1, (size_t)n, fo);
      }
      const int L = len[m];
      if (!no_ac) { memcpy(qb, qs + qpos, (size_t)L); qpos += L; }
      else if_read(&fQ[m], qb, L);
      int64_t end = 0;
      if_read(&fR[m], ob, SZ_READ(L - corlen));
      int lc = 0;
      if (m == 0) { /* :334-343 */
        if_read(&fR[m], &end, sz_meta);
        if (end) {
          for (int i = L - (int)end; i < L - corlen; i++) l[lc++] = (uint8_t)"ACGT"[(ob[i >> 2] >> ((~i & 3) << 1)) & 3];
          for (int i = 0; i < corlen; i++) l[lc++] = (uint8_t)t->pat[core][i];
        }
      }
      for (int i = 0; i < L - (int)end; i++) l[lc++] = (uint8_t)"ACGT"[(ob[i >> 2] >> ((~i & 3) << 1)) & 3];
      for (int i = 0; i < L; i++) { /* :347-354 */
        if (!qb[i]) l[i] = 'N';
        qb[i] = (uint8_t)(qb[i] + phred[m]);
      }
      l[L] = '\n'; qb[L] = '\n';
      fwrite(l, 1, (size_t)L + 1, fo);
      fwrite("+\n", 1, 2, fo);
      fwrite(qb, 1, (size_t)L + 1, fo);
      nameidx++;
    }
    fclose(fo);
    free(qs);
  }
  for (int m = 0; m < nm; m++) { gzclose(fR[m].g); gzclose(fQ[m].g); gzclose(fN[m].g); free(table[m]); }
  return 0;
}
