/*
 * scalce_oracle.h -- CPU restatement of the SCALCE compress/decompress hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under scalce_amd/ (the product) may include,
 * link, dlopen or execute anything in oracle/.  Allowed users: tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() -- as the checker, never as the
 * thing measured or shipped.
 *
 * Every function cites the reference file:line (under /root/reference) whose
 * behaviour it restates.  Pinning status (see DESIGN.md "Oracle"):
 *   - tokenizer + tie-break, 2-bit packer, name codec, bucket order, in-bucket
 *     sort, quality remap + trigram statistics, ac_stat tables, arithmetic
 *     coder/decoder: pinned against the real reference objects built by
 *     oracle/Makefile into oracle/_ref (tests/test_oracle_vs_ref.py) and the
 *     golden vectors under tests/golden/ generated from them.
 *   - quality_mapping_init's sampling loop, AC block framing, .scalce{n,r,q} layout, -B chunk
 *     rule and merge order, decompressor: pinned at FILE level against the whole reference
 *     (oracle/_ref/ref_full = the reference's own compress()/decompress(), every source but
 *     main.cpp compiled where it lies): tests/golden/files.json holds the SHA-256 of what it
 *     wrote for the cases of tests/filecases.py, tests/test_ref_files.py checks orc_cli (CPU)
 *     and the scalce binary (GPU) against them.
 */
#ifndef SCALCE_ORACLE_H
#define SCALCE_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_AC_DEPTH 80                 /* arithmetic.h:47 */
#define ORC_AC_BLOCK (10 * 1024 * 1024) /* arithmetic.cpp:48 */
#define ORC_ROOT_CORE 0x3FFFFFFF        /* MAXBIN-1, const.h:95, reads.cpp:161-164 */

typedef struct orc_trie orc_trie;

/* ---- core table (reads.cpp:253-267, 270-324, 330-410) ---- */
orc_trie *orc_trie_from_bin(const uint8_t *blob, size_t n);   /* patterns.bin layout */
orc_trie *orc_trie_from_text(const char *text, size_t n);     /* -P whitespace list   */
void orc_trie_free(orc_trie *t);
int orc_trie_patterns(const orc_trie *t);
int orc_trie_nodes(const orc_trie *t);          /* trie nodes, root excluded (nodes_count) */
int orc_trie_pattern_len(const orc_trie *t, int pattern);
const char *orc_trie_pattern(const orc_trie *t, int pattern);
/* BFS id (reads.cpp:296) of the node that ends `pattern`; -1 if overwritten */
int orc_trie_pattern_id(const orc_trie *t, int pattern);
void orc_trie_reset_counts(orc_trie *t);        /* bin_size := 0 for every node */

/* ---- per-read functions ---- */
/* aho_search, reads.cpp:413-429.  text = L bases (no newline needed).  Returns
 * bestpos (index of the core's last base) or -1; *pattern = pattern index or -1,
 * *level = core length or 0.  Does NOT bump bin_size. */
int orc_search(const orc_trie *t, const char *text, int L, int *pattern, int *level);
/* aho_trie_bucket's `bin_size++`, reads.cpp:246 (pattern -1 = root) */
void orc_bucket_bump(orc_trie *t, int pattern);
/* output_read, reads.cpp:432-461: rotate around the core and 2-bit pack */
int orc_pack_read(const char *line, int L, int n, int l, uint8_t *dest);
/* output_name, names.cpp:48-62 (no interleave).  name points at '@'. */
int orc_pack_name(const char *name, int use_names, uint8_t *dest);

/* Batched sequential tokenizer: reads = N rows of L bases (row stride `stride`).
 * Applies orc_search + bump in input order (the -T 1 semantics). */
void orc_tokenize_seq(orc_trie *t, const char *reads, int64_t N, int L, int stride,
                      int32_t *pattern_out, int32_t *end_out);

/* aho_output order + bin_prepare/_radix_sort, reads.cpp:466-499,547-634.
 * pattern[i]/end[i] as produced by orc_tokenize_seq; chunk[i] = spill chunk of
 * read i (all 0 for one chunk).  perm_out[k] = input index of the k-th record in
 * final order: bucket id ascending, root last; inside a bucket chunk ascending
 * (compress.cpp:130-159), then suffix-after-core padded with A, stable. */
void orc_bucket_order(const orc_trie *t, const char *reads, int64_t N, int L, int stride,
                      const int32_t *pattern, const int32_t *end, const int32_t *chunk,
                      int64_t *perm_out);

/* ---- qualities (qualities.cpp) ---- */
typedef struct {
  int offset;
  int values[128];
} orc_qmap;
/* quality_mapping_init, qualities.cpp:58-175, from an already-built histogram of
 * the sampled quality characters. */
void orc_qmap_init(orc_qmap *q, const int stat[128], int lossy_percentage);
/* output_quality, qualities.cpp:177-204.  state[0..1] = the two predecessor
 * symbols (500 = none yet), persisting across calls; freq4 = 80^3 counters. */
void orc_quality(const char *qual, const char *read, int L, const orc_qmap *q,
                 uint8_t *dest, uint64_t *freq4, uint32_t state[2], int no_ac);

/* ---- arithmetic coder (arithmetic.cpp) ---- */
/* compress.cpp:296-320: divide by factor, clamp to >= 1, narrow to u32 */
void orc_ac_scale(const uint64_t *freq4, int factor, uint32_t *table_out);
typedef struct {
  uint32_t hi[ORC_AC_DEPTH * ORC_AC_DEPTH][ORC_AC_DEPTH];
  uint32_t lo[ORC_AC_DEPTH * ORC_AC_DEPTH][ORC_AC_DEPTH];
  uint32_t tot[ORC_AC_DEPTH * ORC_AC_DEPTH];
  const uint32_t *cnt;
} orc_acstat;
void orc_acstat_init(orc_acstat *s, const uint32_t *table); /* arithmetic.cpp:54-78 */
/* one block: reset + write + flush (arithmetic.cpp:85-169, 280-287); returns bytes */
size_t orc_ac_encode_block(const orc_acstat *s, const uint8_t *sym, size_t n, uint8_t *out);
/* arithmetic.cpp:173-268, 289-294 */
void orc_ac_decode_block(const orc_acstat *s, const uint8_t *in, size_t nsym, uint8_t *sym);
/* ac_write framing, arithmetic.cpp:318-363: [u32 size][bytes] per 10 MiB block.
 * out must hold n + n/4 + 64 bytes per block worst case; returns bytes written. */
size_t orc_ac_encode_stream(const orc_acstat *s, const uint8_t *sym, size_t n, uint8_t *out,
                            size_t out_cap, int threads);

/* ---- whole pipelines (compress.cpp:721-848, decompress.cpp:79-397) ---- */
typedef struct {
  int paired;                /* -r */
  int use_names;             /* !-n */
  const char *library;       /* -n value */
  int no_ac;                 /* -A */
  int lossy;                 /* -p */
  int sample;                /* -s, default 100000 */
  int gz;                    /* -c gz (1) or -c no (0) */
  uint64_t bucket_set_size;  /* -B in bytes, default 4 GiB */
  int threads;               /* AC block fan-out only; order is always -T 1 */
  int verbose;
} orc_opts;
void orc_opts_default(orc_opts *o);
/* returns 0 on success; error text on stderr like the reference's ERROR() */
int orc_compress_files(orc_trie *t, const char *fastq1, const char *out_prefix, const orc_opts *o);
int orc_decompress_files(const orc_trie *t, const char *scalce_path, const char *out_prefix,
                         const orc_opts *o);

#ifdef __cplusplus
}
#endif
#endif
