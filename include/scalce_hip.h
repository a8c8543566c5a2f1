/*
 * scalce_hip.h -- C ABI of the MI355X (gfx950) SCALCE hot path.
 *
 * The reference (sfu-compbio/scalce 2.8) has no plugin / FFI seam: it is one executable with
 * global state, and its hot path is a set of per-read calls made from thread()
 * (compress.cpp:673-706) and per-stream calls made from the final writer
 * (compress.cpp:296-335,380-391,411-412).  Per-read signatures cannot be driven from a GPU,
 * so every entry point below is the BATCHED equivalent of the reference functions it names.
 * Plain C types, caller-visible integer status, no global state, one context per device.
 *
 * Pointers whose name starts with d_ are DEVICE pointers (HBM); `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  All calls are asynchronous on `stream`
 * unless stated otherwise; scalce_batch_finish() synchronises and reports device-side errors.
 */
#ifndef SCALCE_HIP_H
#define SCALCE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCALCE_OK 0
#define SCALCE_ERR_ARG 1       /* bad argument / state */
#define SCALCE_ERR_HIP 2       /* HIP runtime failure (message in scalce_last_error) */
#define SCALCE_ERR_FORMAT 3    /* malformed input: the cases where the reference prints (ERROR) and exits */
#define SCALCE_ERR_CAPACITY 4  /* batch larger than the capacity it was created with */
/* (5 was SCALCE_ERR_UNCUT in rounds 1-3: a run that -B does not cut is one spill chunk and goes to rank 0 as a whole) */

#define SCALCE_AC_DEPTH 80                  /* arithmetic.h:47 */
#define SCALCE_AC_BLOCK (10 * 1024 * 1024)  /* arithmetic.cpp:48 */
#define SCALCE_ROOT_CORE 0x3FFFFFFF         /* MAXBIN-1: const.h:95, reads.cpp:161-164 */

typedef struct scalce_ctx scalce_ctx;
typedef struct scalce_batch scalce_batch;

/* ---- context ---------------------------------------------------------------------------- */
int scalce_ctx_create(int device, scalce_ctx **out);
void scalce_ctx_destroy(scalce_ctx *ctx);
const char *scalce_last_error(const scalce_ctx *ctx);

/* ---- core table: read_patterns (reads.cpp:330-377), read_patterns_from_file (:379-410),
 *      pattern_insert (:253-267), prepare_aho_automata (:270-324).  Built on the host,
 *      uploaded as a BFS-ordered DFA (state id == the reference's BFS `id`, root 0). ------- */
int scalce_patterns_load_bin(scalce_ctx *ctx, const void *blob, size_t nbytes);
int scalce_patterns_load_text(scalce_ctx *ctx, const char *text, size_t nbytes);
int scalce_patterns_count(const scalce_ctx *ctx);   /* cores in file order (the index .scalcer stores) */
int scalce_patterns_states(const scalce_ctx *ctx);  /* automaton states, root included */
int scalce_patterns_buckets(const scalce_ctx *ctx); /* distinct cores = buckets, root excluded */
/* core string / length by file-order index (decompress.cpp:269,341 use patterns[core]) */
int scalce_pattern_length(const scalce_ctx *ctx, int pattern);
const char *scalce_pattern_string(const scalce_ctx *ctx, int pattern);

/* Host-only description of the table builder (runs without a device): bucket_pattern_out[k] = file-order
 * index of the core whose bucket is emitted k-th (aho_output order, reads.cpp:466-499), root (0x3FFFFFFF) last. */
int scalce_patterns_describe_host(const void *blob, size_t nbytes, int is_text, int32_t *bucket_pattern_out,
                                  size_t cap, int32_t *n_states, int32_t *n_buckets);

/* ---- quality model: quality_mapping_init (qualities.cpp:58-175), host only --------------- */
typedef struct {
  int32_t offset;      /* 33 or 64 */
  int32_t values[128]; /* replacement table */
} scalce_qmap;
/* stat[c] = occurrences of quality character c in the sampled records */
void scalce_qmap_init(scalce_qmap *q, const int32_t stat[128], int lossy_percentage);

/* ---- batch: one FASTQ shard resident in HBM --------------------------------------------- */
typedef struct {
  int32_t read_len[2];       /* bases per read, mate 1 / mate 2 (read_length[], compress.cpp:61) */
  int32_t paired;            /* -r */
  int32_t use_names;         /* 0 under -n */
  int32_t no_ac;             /* -A */
  scalce_qmap qmap[2];
  uint64_t bucket_set_size;  /* -B in bytes (spill rule, compress.cpp:702-715); 0 = never spill */
  /* state carried in from earlier shards of the same run (multi-GPU / multi-file); NULL = none */
  uint32_t qprev[2][2];      /* last two quality symbols before this shard, 500 = none (qualities.cpp:179) */
} scalce_params;

void scalce_params_default(scalce_params *p);
/* Allocates every device buffer a shard of up to max_reads records / max_text bytes per mate
 * needs.  Synchronous. */
int scalce_batch_create(scalce_ctx *ctx, const scalce_params *p, uint64_t max_reads, uint64_t max_text,
                        scalce_batch **out);
void scalce_batch_destroy(scalce_batch *b);
/* Batches that share their front-stage buffers.  Rows, tokens, tie-break events and sort scratch are dead once a shard is
 * emitted -- the coder reads the reordered stream and writes the coded one -- so batches whose front stages (ingest .. emit)
 * run one after the other on ONE stream can share a single set: 15 GB instead of 35 GB of HBM per shard in flight at
 * 50 M x 100 bp.  Outputs 5, 6, 9, 10 of a batch (tokens, permutation, q' and name lengths in input order) are only valid
 * until the next batch of the workspace starts its front stages.  Destroy the batches before the workspace. */
typedef struct scalce_workspace scalce_workspace;
int scalce_workspace_create(scalce_ctx *ctx, scalce_workspace **out);
void scalce_workspace_destroy(scalce_workspace *w);
int scalce_batch_create_shared(scalce_ctx *ctx, const scalce_params *p, uint64_t max_reads, uint64_t max_text,
                               scalce_workspace *w, scalce_batch **out);

/* record reader of thread() (compress.cpp:614-666): newline index of the FASTQ text of one
 * mate, validation of the fixed read length, 2-bit packing of the bases (getval, const.cpp:47),
 * name slicing (output_name, names.cpp:48-62).  The text is not read again once the call has returned and `stream`
 * has passed it. */
int scalce_batch_ingest(scalce_batch *b, int mate, const uint8_t *d_text, uint64_t nbytes, void *stream);
/* The record reader for inputs that do not fit HBM as text (the reference reads records one by one from files of any
 * size and spills buckets to disk, compress.cpp:614-717; here the TEXT is streamed and what is derived from it stays
 * resident: 2-bit rows, q', names, tokens -- about two thirds of the text).  scalce_batch_append takes the next piece of the
 * read stream BEHIND the rows the batch already holds: as many complete records as both mates' pieces contain (the
 * mates are read in step), consumed[m] = bytes of piece m that were used -- the caller puts the rest in front of its
 * next piece; final_piece = 1 makes a leftover an error (line count not a multiple of 4, mates of different length).
 * Ingest, quality counters (prev[] runs across pieces, qualities.cpp:179) and the tie-break of the new rows against ALL
 * rows before them (bin_size is cumulative, reads.cpp:246) are done when the call returns and the text may be
 * overwritten; scalce_batch_order / _emit / _entropy then run once over everything.  d_text* 16-byte aligned, at most
 * max_text bytes each; the row arrays grow beyond max_reads when they must.  The first append after create, reset or a
 * one-piece ingest starts a new run.  Synchronises `stream`.
 * flags: SCALCE_APPEND_FINAL = the final piece; SCALCE_APPEND_NO_QUALITY leaves the trigram counters alone (the rows were
 * counted elsewhere: sharded runs); SCALCE_APPEND_NO_TOKENIZE defers the tie-break -- scalce_batch_tokenize / _begin then
 * cover every row not tokenized yet (sharded runs resolve it across ranks once all rows are in place). */
#define SCALCE_APPEND_FINAL 1
#define SCALCE_APPEND_NO_QUALITY 2
#define SCALCE_APPEND_NO_TOKENIZE 4
int scalce_batch_append(scalce_batch *b, const uint8_t *d_text1, uint64_t n1, const uint8_t *d_text2, uint64_t n2,
                        int flags, uint64_t consumed[2], void *stream);
int scalce_batch_reset(scalce_batch *b);

/* The streaming host around scalce_batch_append -- what replaces the reader half of thread() (compress.cpp:614-671) and
 * the spill files (:708-715) for inputs of any size: one reader thread per mate pulls the stream through `rd` into
 * pinned chunks of piece_bytes (0 = 1 GiB), the chunks go up with hipMemcpyAsync while the previous piece is ingested,
 * counted and tokenized, the unconsumed tail of a piece is put in front of the next one on the device; then order, emit
 * and entropy run once over the run.  On success *out holds the results (scalce_batch_output; the caller destroys it).
 * rd(user, dst, cap) returns the bytes it stored (any number up to cap), 0 at the end of the stream, < 0 on error; it is
 * called from the reader thread of its mate only.  reads_hint sizes the row arrays (0: they grow as the run comes in).
 * flags: SCALCE_STREAM_LEAN releases device buffers as stages finish (scalce_batch_set_lean); SCALCE_STREAM_DEFER_ENTROPY
 * returns behind the emit stage -- the caller runs scalce_batch_entropy_begin on a stream of its own and fetches the
 * read and name streams while the coder works.  errbuf receives the message on failure. */
#define SCALCE_STREAM_LEAN 1
#define SCALCE_STREAM_DEFER_ENTROPY 2
typedef int64_t (*scalce_read_fn)(void *user, void *dst, uint64_t cap);
typedef struct {
  double total_s, read_wait_s, h2d_wait_s, front_s, order_s, emit_s, entropy_s;
  uint64_t rounds, reads, bytes[2];
} scalce_stream_stats;
int scalce_stream_compress(scalce_ctx *ctx, const scalce_params *p, scalce_read_fn rd1, void *user1, scalce_read_fn rd2,
                           void *user2, uint64_t piece_bytes, uint64_t reads_hint, int flags, scalce_batch **out,
                           scalce_stream_stats *stats, char *errbuf, size_t errcap);
/* lean = 1: a stage releases the device buffers that no later stage reads (q' in input order once the reordered stream
 * exists, rows and sort scratch once the records are emitted): outputs 5, 6, 9 become unavailable, runs sized for most
 * of HBM fit. */
void scalce_batch_set_lean(scalce_batch *b, int lean);
/* Row layout of the batch.  A single-end batch with names keeps ONE row per read -- q' | name cell | packed bases, what the
 * reference keeps per read in its bucket blob (reads.h:52-58, compress.cpp:675-706) -- so that the emit stage reaches
 * everything it reorders through one random access.  on = 0 (before anything is ingested) goes back to separate arrays whose
 * rows lie back to back: for callers that read SCALCE_OUT_QINPUT as one array (sharded runs); lean batches do so by themselves.
 * Returns SCALCE_ERR_ARG when the batch cannot take the layout asked for. */
int scalce_batch_set_fused_rows(scalce_batch *b, int on);
/* Where the reordered q' stream of the emit stage lives (arithmetic.cpp:318-363 cuts it into 10 MiB blocks).  on = 1: in the
 * workspace the batch shares with others instead of in the batch itself -- for callers that pass the stream on right behind
 * the emit stage and never code it where it lies (scalce_sharded_compress: every rank hands its stream to the owners of the
 * run-wide block ranges): SCALCE_OUT_QSTREAM is then valid until the next batch of the workspace runs its emit stage, and the
 * batch holds 5 GB less per 50 M reads of 100 bp.  Not with -A (the stream is the output) or lean batches: SCALCE_ERR_ARG. */
int scalce_batch_set_stream_scratch(scalce_batch *b, int on);
/* Coding in place.  on = 1: in grouped launches (scalce_batch_entropy_begin_group) the coder writes block k's bytes over block
 * k's own symbols -- the output of a block lags its input (arithmetic.cpp:85-169 emits fewer bits than it reads symbols' worth)
 * -- and the batch holds no block buffers (3.2 GB less per 50 M reads of 100 bp; SCALCE_OUT_QSTREAM is gone once the shard is
 * coded).  A block whose output would catch up with its input cannot be coded again from symbols that are already
 * overwritten: the shard is then run again from its TEXT with buffers of its own when it is collected, so the caller keeps
 * the text passed to scalce_batch_front / _ingest where it is until the shard has been collected.  Not with -A or lean
 * batches (SCALCE_ERR_ARG); a stream handed in by the caller (sharded runs) is never coded in place. */
int scalce_batch_set_code_in_place(scalce_batch *b, int on);
uint64_t scalce_batch_reruns(const scalce_batch *b);  /* shards of this batch that had to be run again from their text */
/* edge[0..1] = the first two, edge[2..3] = the last two q' symbols of the rows held (input order), *nsym = how many there are,
 * *read_len (may be NULL) = symbols per row: what a rank of a sharded run tells its neighbours (qualities.cpp:179-198: prev[]
 * runs across reads, so two trigrams straddle every rank boundary).  Runs on `stream` and synchronises it. */
int scalce_batch_qinput_edges(scalce_batch *b, int mate, uint8_t edge[4], uint64_t *nsym, int32_t *read_len, void *stream);
/* output_quality (qualities.cpp:177-204): q' = map[q]-offset (N -> 0) and the order-2 trigram
 * counters ac_freq4 over the input-order stream of this shard. */
int scalce_batch_quality(scalce_batch *b, void *stream);
/* aho_search (reads.cpp:413-429) for every read, including the cumulative bin_size tie-break
 * resolved exactly in input order (-T 1 semantics).  d_prior_counts: per-bucket counts of reads
 * assigned by EARLIER shards (bucket order = scalce_bucket_*), or NULL. */
int scalce_batch_tokenize(scalce_batch *b, const uint64_t *d_prior_counts, void *stream);
/* The same in two halves: _begin (both walks, candidates, events, first counts) and _settle (below).  A run sharded over
 * several GPUs calls _begin on every rank at once and _settle rank by rank: rank r settles against the counts of ranks
 * 0 .. r-1 (d_prior_counts = their sum per bucket), then hands its own SCALCE_OUT_BUCKET_COUNTS on (scalce_sharded_compress). */
int scalce_batch_tokenize_begin(scalce_batch *b, void *stream);
int scalce_batch_tokenize_end(scalce_batch *b, void *stream);
/* scalce_batch_tokenize = _begin + _settle: _settle resolves the tie-break of this batch on its own against fixed prior
 * counts (window by window in input order) and ends the tokenization.  Split so that a caller can enqueue other work of the
 * shard on a second stream beside the tie-break's many small launches. */
int scalce_batch_tokenize_settle(scalce_batch *b, const uint64_t *d_prior_counts, void *stream);
/* Sharded runs: the cuts the -B rule makes inside this batch's rows when `carry_in` bytes of records are already in the
 * chunk that is open where they begin (the rows of the ranks before): cuts_host[i] = row in front of which chunk i + 1
 * begins (1 .. N; at most cap), carry_out = bytes in the chunk still open behind the last row.  Needs the rows ingested,
 * not tokenized (record sizes depend on the core's length only). */
int scalce_batch_chunk_plan(scalce_batch *b, uint64_t carry_in, uint64_t *cuts_host, uint32_t cap, uint32_t *ncuts,
                            uint64_t *carry_out, void *stream);
/* Byte offset, in the text of the piece ingested last, at which record `row` (0 .. rows of that piece) begins; runs on
 * `stream`, behind the ingest of that piece. */
int scalce_batch_text_offset(scalce_batch *b, int mate, uint64_t row, uint64_t *offset, void *stream);
/* Sharded runs: the row range a rank holds changes at both ends once the run-wide spill-chunk cuts are known (rank boundaries
 * move to the nearest cut, compress.cpp:702-715): rows [keep_first, keep_first + keep_rows) of the batch stay as they are, the
 * records of `front` (FASTQ text on the device, whole records, 16-byte aligned; [mate]) become rows in front of them, those of
 * `back` rows behind them.  Only the records that arrive are ingested and walked; the quality statistics are not touched (every
 * record was counted by the rank that ingested it first).  Before any tokenization of the batch. */
int scalce_batch_rewindow(scalce_batch *b, uint64_t keep_first, uint64_t keep_rows, const uint8_t *const front[2],
                          const uint64_t front_bytes[2], const uint8_t *const back[2], const uint64_t back_bytes[2], void *stream);
/* Spill-chunk boundaries given by the caller instead of the -B rule: starts[0] = 0 < starts[1] < ...;
 * records of chunk i precede those of chunk i+1 inside every bucket (merge order, compress.cpp:104-159).
 * A sharded run uses one chunk per shard. */
int scalce_batch_set_chunks(scalce_batch *b, const uint64_t *starts, uint32_t n);
/* aho_trie_bucket + aho_output + bin_prepare/_radix_sort (reads.cpp:233-250,466-499,547-634)
 * + the merge order of spilled chunks (compress.cpp:104-159): the output permutation. */
int scalce_batch_order(scalce_batch *b, void *stream);
/* output_read (reads.cpp:432-461) + bin_dump (:91-180) + per-bucket headers
 * (compress.cpp:364-379): the .scalcer / .scalcen payloads and the reordered quality stream. */
int scalce_batch_emit(scalce_batch *b, void *stream);
/* table scaling (compress.cpp:296-320), ac_stat (arithmetic.cpp:54-78) and ac_write /
 * ac_coder (:85-169,318-363): [u32 size][bytes] per 10 MiB block of the reordered stream.
 * d_table_override (512000 x u32 per mate, already scaled) replaces this shard's own
 * statistics when the run-wide table was reduced across shards; NULL = use own. */
int scalce_batch_entropy(scalce_batch *b, const uint32_t *d_table_override, void *stream);
/* The same stage in two halves, for callers that keep several shards in flight: _begin only enqueues (table,
 * coder, framing -- one short wait for the table's largest context total, none for the coder), so the caller
 * can run the front stages of the next shard on another stream while the coder -- a long kernel that leaves
 * most of the chip idle, one wavefront per 10 MiB block -- works in the background; _end (or
 * scalce_batch_finish) waits for it and fills in the size of SCALCE_OUT_QUAL.  This is the device-side
 * counterpart of the reference coding `-T` blocks per batch on its thread pool (arithmetic.cpp:349-357). */
int scalce_batch_entropy_begin(scalce_batch *b, const uint32_t *d_table_override, void *stream);
int scalce_batch_entropy_end(scalce_batch *b, void *stream);
/* The begin half for several shards at once: ONE coder launch over the blocks of all of them, with several blocks per
 * chain wavefront (four: 0.57 x the SIMD time per block of the one-block kernel at 1.13 x its latency; eight: ~0.35 x
 * at 1.3 x).  The library picks the fewest blocks per wave that keep the launch at one workgroup per CU -- two
 * 50 M-read shards are 954 blocks = 239 workgroups of four, three shards 179 workgroups of eight -- so every coder wave
 * has a SIMD to itself by construction and the group is coded in little more than the time one shard takes.  Tables
 * are prepared on prep_stream (the host waits there, never behind a running coder), the coder and the framing are
 * enqueued on stream.  Each shard is completed by its own scalce_batch_entropy_end / scalce_batch_finish (on any
 * stream, after `stream` has reached the end of the launch). */
int scalce_batch_entropy_begin_group(scalce_batch **batches, int n, void *prep_stream, void *stream);
/* The same for the LAST launch of a run (last == 1: nothing will be queued behind it): picks the coder kernel by how soon
 * the launch is done instead of by how few CUs it holds beside the next shards' front stages.  last == 2: a launch of one
 * or two shards at the START of a run, with more shards on their way: the kernel that holds the fewest CUs whatever the
 * size of the launch.  last == 3: one of the last launches of a run, a front stage or two still to come: a kernel between
 * the two (eight blocks per chain wave for up to 1024 blocks).  Same bytes. */
int scalce_batch_entropy_begin_group_last(scalce_batch **batches, int n, void *prep_stream, void *stream, int last);
/* Sharded runs: code a caller-assembled range of the run-wide reordered stream (it must start on a 10 MiB
 * block boundary) against the run-wide table; result in SCALCE_OUT_QUAL of `mate`. */
int scalce_batch_entropy_stream(scalce_batch *b, int mate, const uint32_t *d_table, const uint8_t *d_symbols,
                                uint64_t nsym, void *stream);
int scalce_batch_entropy_stream_begin(scalce_batch *b, int mate, const uint32_t *d_table, const uint8_t *d_symbols,
                                      uint64_t nsym, void *stream);
/* ... or only remembered, to be coded by the next scalce_batch_entropy_begin_group that includes this shard. */
int scalce_batch_entropy_stream_prepare(scalce_batch *b, int mate, const uint32_t *d_table, const uint8_t *d_symbols,
                                        uint64_t nsym, void *stream);
/* Table scaling on its own (compress.cpp:297-313): d_table[i] = max(1, (1 + d_counters[i]) / factor) for the 512000
 * counters of one mate (SCALCE_OUT_FREQ4 layout: raw trigram counts; the reference's counters all start at 1,
 * qualities.cpp:191-196).  factor = 1 + symbols / (2^32 - 1) over the RUN (:297-303). */
int scalce_ac_scale(scalce_ctx *ctx, const uint64_t *d_counters, uint32_t factor, uint32_t *d_table, void *stream);
/* Piecewise device copy: dst[piece_dst[p] + i] = src[piece_src[p] + i]; pieces contiguous in src, sorted. */
int scalce_copy_pieces(scalce_ctx *ctx, const uint8_t *d_src, uint8_t *d_dst, const uint64_t *d_piece_src,
                       const uint64_t *d_piece_dst, uint32_t npieces, uint64_t total_bytes, void *stream);
/* all of the above in order */
int scalce_batch_compress(scalce_batch *b, const uint8_t *d_text1, uint64_t n1, const uint8_t *d_text2,
                          uint64_t n2, void *stream);
/* Every stage in front of the entropy coder (ingest .. emit) of a shard resident as text (mate 2: NULL / 0 for single-end). */
int scalce_batch_front(scalce_batch *b, const uint8_t *d_text1, uint64_t n1, const uint8_t *d_text2, uint64_t n2, void *stream);
/* Synchronises `stream`, checks the device error word, fills the host-side result sizes. */
int scalce_batch_finish(scalce_batch *b, void *stream);

/* results (device pointers owned by the batch; sizes valid after scalce_batch_finish) */
enum {
  SCALCE_OUT_READS = 0,     /* .scalcer payload after the 16-byte header, mate m            */
  SCALCE_OUT_NAMES = 1,     /* .scalcen payload after magic+flag (empty under -n), mate 1    */
  SCALCE_OUT_QUAL = 2,      /* .scalceq payload after header+table+total: AC blocks, or raw q' under -A */
  SCALCE_OUT_TABLE = 3,     /* 512000 x u32 scaled frequency table (file order), mate m      */
  SCALCE_OUT_FREQ4 = 4,     /* 512000 x u64 raw trigram counters of this shard, mate m       */
  SCALCE_OUT_TOKENS = 5,    /* N x {int32 pattern (file order, -1 = none), int32 end}, input order */
  SCALCE_OUT_PERM = 6,      /* N x u32: input index of the k-th emitted record               */
  SCALCE_OUT_QSTREAM = 7,   /* N*L bytes: reordered q' stream, mate m                         */
  SCALCE_OUT_BUCKET_COUNTS = 8, /* (buckets+1) x u64 reads per bucket, emission order, root last */
  SCALCE_OUT_QINPUT = 9,    /* N*L bytes: q' in input order, mate m                          */
  SCALCE_OUT_NAMELEN = 10,  /* N bytes: stored name length per read, input order             */
  SCALCE_OUT_BUCKET_NAME_BYTES = 11 /* (buckets+1) x u64: bytes of each bucket's records in SCALCE_OUT_NAMES (after emit) */
};
int scalce_batch_output(const scalce_batch *b, int which, int mate, const void **d_ptr, uint64_t *nbytes);
/* Framing on the way out (replaces the copy loop of ac_write, arithmetic.cpp:318-363: the reference frames each coded
 * block -- [u32 size][bytes] -- as it writes it to the file).  With frame_on_demand set, the entropy stage leaves the coded
 * blocks where the coder wrote them and only lays the frames out; scalce_batch_qual_window then delivers bytes
 * [offset, offset + nbytes) of SCALCE_OUT_QUAL into dst -- device memory, or pinned host memory (4-byte aligned): the
 * framing kernel's stores are the download -- so the stream never exists a second time in HBM.  Valid after
 * scalce_batch_finish / _entropy_end.  A caller that asks scalce_batch_output for SCALCE_OUT_QUAL still gets the whole
 * stream as one device buffer; it is put together at that moment. */
int scalce_batch_set_frame_on_demand(scalce_batch *b, int on);
int scalce_batch_qual_bytes(const scalce_batch *b, int mate, uint64_t *nbytes);  /* size of SCALCE_OUT_QUAL, nothing put together */
int scalce_batch_qual_window(scalce_batch *b, int mate, uint64_t offset, uint64_t nbytes, void *dst, void *stream);
uint64_t scalce_batch_reads(const scalce_batch *b);
int scalce_batch_params(const scalce_batch *b, scalce_params *out);  /* the parameters it was created with */
/* measurement hook for bench.py: accumulated device time (ms, hipEvent) and launch count of
 * stage `which` (0 ingest,1 quality,2 tokenize,3 order,4 emit,5 entropy) since the last reset */
int scalce_batch_stage_ms(scalce_batch *b, int which, float *ms, int *launches);
void scalce_batch_stage_reset(scalce_batch *b, int enable);

/* HIP-event pairs around every launch of the dominant kernel (ac_encode_k) on the caller's stream.
 * kernel_timing(1) arms and clears; kernel_ms synchronises the recorded events and returns the summed
 * duration, the launch count and the algorithmic bytes (symbols read, coded bytes written). */
void scalce_batch_kernel_timing(scalce_batch *b, int enable);
int scalce_batch_kernel_ms(scalce_batch *b, double *total_ms, int *launches, uint64_t *bytes_in, uint64_t *bytes_out);

/* plumbing for hosts without a HIP binding of their own (ctypes, cgo): blocking copies */
int scalce_memcpy_d2h(scalce_ctx *ctx, void *dst_host, const void *src_dev, uint64_t nbytes);
int scalce_memcpy_h2d(scalce_ctx *ctx, void *dst_dev, const void *src_host, uint64_t nbytes);
int scalce_memcpy_d2d(scalce_ctx *ctx, void *dst_dev, const void *src_dev, uint64_t nbytes, void *stream); /* async */
/* diagnostics of the last run: tie reads, candidate events, fixed-point sweeps, spill chunks, records that
 * needed the second sort phase, 1 if the tie-break ended in the sequential fallback (tie_sequential_k) */
int scalce_batch_stats(const scalce_batch *b, uint32_t out[6]);

/* Device self-test of the arithmetic coder's closed-form step (multiply-high by reciprocal fractions, merged
 * renormalisation shift) against the literal loop of arithmetic.cpp:122-152 on `ncases` random and crafted
 * states.  general = 0: the production step on well-formed states with totals <= 2^30; general = 1: the
 * all-states step, including inverted intervals; general = 2: the plain-round step on the (lo, range) state with
 * its fall-back, as the encoder's inner loop runs it.  out[0] = mismatches, out[1..5] = lo, hi, c_lo, c_hi, total of the first one. */
int scalce_selftest_ac(scalce_ctx *ctx, uint64_t ncases, uint32_t seed, int general, uint32_t out[6]);

/* ---- decode side (next row of SURVEY 8f-1): ac_read / ac_decoder (arithmetic.cpp:173-268,
 *      365-400).  d_blocks = [u32 size][bytes]... as written by scalce_batch_entropy. ---------- */
int scalce_ac_decode(scalce_ctx *ctx, const uint32_t *table_host, const uint8_t *d_blocks, uint64_t nbytes,
                     uint64_t nsymbols, uint8_t *d_symbols_out, void *stream);

/* Records back to FASTQ text on the device: the per-record body of decompress.cpp:240-366 -- bucket directory
 * (:262-270), un-rotation of the 2-bit bases around the core (:331-345), N where the quality is 0 (:350-351), name
 * line (:290-299; library mode "@<library>.<index>" :300-304), '+' line, qualities + phred offset.
 *   reads_host   .scalcer payload behind its header (magic, no_ac, read length), host memory: the directory walk
 *                is serial; has_buckets = 1 for mate 1 (bucket headers, end metadata), 0 for mate 2 (bare records,
 *                no core -- the reference's stale `corlen` of decompress.cpp:250,332 is NOT reproduced);
 *   d_qual       nrecords * read_len quality symbols in archive order (scalce_ac_decode's output, or the raw
 *                bytes of a -A archive), device memory;
 *   names_host   .scalcen payload behind magic and use_names byte, or NULL for library mode (`library` used);
 *   mate_digit   0, or '1' / '2' for paired archives: a name ending in "/x" gets this digit (:296-298);
 *   d_out        device buffer of out_cap >= scalce_fastq_text_bytes(...) bytes;
 *   record_offsets_host  optional, nrecords + 1 entries: where each record starts in the text (-S splitting).
 * Returns when the text is complete. */
uint64_t scalce_fastq_text_bytes(int read_len, uint64_t nrecords, uint64_t names_bytes, const char *library /* NULL: names */);
int scalce_fastq_records(scalce_ctx *ctx, int read_len, int has_buckets, const uint8_t *reads_host, uint64_t reads_bytes,
                         uint64_t nrecords, const uint8_t *d_qual, int64_t phred_offset, const uint8_t *names_host,
                         uint64_t names_bytes, const char *library, int mate_digit, uint8_t *d_out, uint64_t out_cap,
                         uint64_t *out_bytes, uint64_t *record_offsets_host, void *stream);

/* ---- runs sharded over several GPUs: one process per GPU, ONE archive -----------------------------------------
 * The reference has no distributed mode; what it carries across reads is what ranks exchange (see comm.cpp).  The
 * archive of a sharded run is byte for byte the archive of the same input on one GPU -- and of the reference at -T 1 with
 * the same -B -- for any number of ranks: rank boundaries are moved to the nearest spill-chunk boundary of the run-wide -B
 * rule (records change owner as text, once), so that "rank-major inside a bucket" IS the merge order of compress.cpp:104-159.
 * -B must be set (the reference's default is 4G) and must cut the run at least once; ranks whose share is smaller than a
 * chunk may end up without records. */
typedef struct scalce_comm scalce_comm;
#define SCALCE_COMM_ID_BYTES 128
int scalce_comm_unique_id(uint8_t id[SCALCE_COMM_ID_BYTES]);  /* rank 0 makes it, the launcher hands it to every rank */
int scalce_comm_create_rccl(int device, int world, int rank, const uint8_t id[SCALCE_COMM_ID_BYTES], scalce_comm **out);
/* rehearsal transport: `world` processes sharing one GPU, staged through POSIX shared memory `name` (slot_bytes per rank,
 * 0 = 64 MiB: the largest message) */
int scalce_comm_create_shm(int device, int world, int rank, const char *name, uint64_t slot_bytes, scalce_comm **out);
void scalce_comm_destroy(scalce_comm *c);
const char *scalce_comm_error(const scalce_comm *c);
int scalce_comm_world(const scalce_comm *c);
int scalce_comm_rank(const scalce_comm *c);
/* Largest message handed to the transport in one piece (default 1 GiB: RCCL loses the tail of larger ones between a rank and
 * itself, comm.cpp); tools/rccl_big_send.py raises it to show where. */
void scalce_comm_set_piece_bytes(scalce_comm *c, uint64_t bytes);
int scalce_comm_barrier(scalce_comm *c, void *stream);
int scalce_comm_all_gather(scalce_comm *c, const void *d_send, void *d_recv, uint64_t bytes_per_rank, void *stream);
int scalce_comm_all_reduce_sum_u64(scalce_comm *c, uint64_t *d_buf, uint64_t count, void *stream);
/* One message between two ranks (what a chain of ranks hands on: the counts of the tie-break, rank r to rank r + 1). */
int scalce_comm_send(scalce_comm *c, const void *d_buf, uint64_t bytes, int peer, void *stream);
int scalce_comm_recv(scalce_comm *c, void *d_buf, uint64_t bytes, int peer, void *stream);
int scalce_comm_all_to_all_v(scalce_comm *c, const void *d_send, const uint64_t *send_bytes, void *d_recv,
                             const uint64_t *recv_bytes, void *stream);
/* The same with explicit offsets: the bytes for rank d start at d_send + send_off[d], those from rank src land at
 * d_recv + recv_off[src] -- a rank sends ranges of a buffer where they lie and keeps its own part out of the transport. */
int scalce_comm_all_to_all_vo(scalce_comm *c, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes,
                              void *d_recv, const uint64_t *recv_off, const uint64_t *recv_bytes, void *stream);

typedef struct {
  int32_t world, rank;
  uint32_t nb1;              /* buckets + 1 (root last) */
  uint32_t rounds;           /* collective rounds of the tie-break */
  uint32_t sweeps;           /* local sweeps of this rank */
  uint32_t chunks_total;     /* spill chunks of the run */
  uint64_t reads_total;      /* of the run */
  uint64_t first_read;       /* run-wide index of this rank's first record AFTER the boundaries moved */
  uint64_t reads_local;      /* records this rank holds after the move */
  uint64_t moved_in[2];      /* records received from the rank before / behind */
  uint64_t *counts;          /* [world][nb1] reads per (rank, bucket), emission order */
  uint64_t *name_bytes;      /* [world][nb1] bytes of their name records */
  uint64_t sym_lo[2], sym_hi[2];  /* per mate: this rank's range of the run-wide reordered quality stream (whole 10 MiB blocks) */
  uint64_t coded_bytes[2][64];    /* per mate and rank: bytes of framed blocks (0 with SCALCE_SHARD_PREPARE_ONLY) */
  void *keep[8];             /* device buffers the coder still reads (freed by scalce_shard_result_free) */
  uint64_t keep_bytes[8];
  uint32_t magic;            /* set by the library: a result handed back in keeps its buffers */
} scalce_shard_result;
#define SCALCE_SHARD_PREPARE_ONLY 1  /* hand the rank's block range to the batch (scalce_batch_entropy_stream_prepare): the
                                        caller codes several shards with one scalce_batch_entropy_begin_group */
#define SCALCE_SHARD_CODER_ASYNC 2   /* only enqueue the coder on coder_stream (scalce_batch_entropy_stream_begin) */
/* SPMD: every rank calls it with its own contiguous piece of the read stream (rank order = input order), resident in
 * HBM.  On return the batch holds this rank's pieces of the archive: SCALCE_OUT_READS / _NAMES = its records per bucket
 * (bucket b of the archive = header, then rank 0's records of b, rank 1's, ...: result->counts / name_bytes say where),
 * SCALCE_OUT_QUAL = the framed blocks of sym_lo .. sym_hi, SCALCE_OUT_TABLE = the run-wide table.  `b` must have been
 * created with bucket_set_size != 0 and default qprev.  `result` must be zeroed before its first use; handing the result
 * of an earlier call on the same batch back in reuses its device buffers (no allocation, hence no device-wide
 * synchronisation, in the steady state of a pipeline). */
int scalce_sharded_compress(scalce_comm *comm, scalce_ctx *ctx, scalce_batch *b, const uint8_t *d_text1, uint64_t n1,
                            const uint8_t *d_text2, uint64_t n2, int flags, void *stream, void *coder_stream,
                            scalce_shard_result *result);
void scalce_shard_result_free(scalce_shard_result *r);
/* The host-side plan math of a sharded run on its own (no device needed; see sharded.cpp): rank boundaries g[0..world]
 * moved to the nearest cut of the run-wide -B rule, and the deal of the run-wide reordered quality stream in contiguous
 * ranges of whole 10 MiB blocks. */
int scalce_shard_plan_boundaries(int world, const uint64_t *g, const uint64_t *cuts_sorted, uint64_t ncuts, uint64_t *gn);
int scalce_shard_plan_blocks(int world, int rank, uint32_t nb1, const uint64_t *counts /*[world][nb1]*/, uint64_t read_len,
                             uint64_t *send_bytes, uint64_t *recv_bytes, uint64_t *lo, uint64_t *hi, uint64_t *piece_src,
                             uint64_t *piece_dst, uint64_t *npieces);

/* ---- shards in flight (pipeline.cpp) -----------------------------------------------------------------------------
 * The reference keeps its cores busy by handing -T blocks of a batch to coder threads while the reader goes on
 * (arithmetic.cpp:349-357, compress.cpp:781-786).  The device-side counterpart for a caller with MANY shards: `nslots`
 * batches (usually sharing one scalce_workspace), the front stages (scalce_batch_ingest .. _emit) of the next shards on one
 * stream beside the coder launches of the previous ones on `coder_streams` others, `group` shards per coder launch
 * (nslots >= 2 * group for launches that go out as they fill up), a shard retired on an event behind its launch.
 * external_coder != 0: the caller enqueues or prepares the coder itself (scalce_sharded_compress with
 * SCALCE_SHARD_CODER_ASYNC / _PREPARE_ONLY).
 *     for every shard:  scalce_pipeline_acquire(p, &slot, &retired);   (retired: take the outputs of the shard that was there)
 *                       ... front stages of the shard into batches[slot] on scalce_pipeline_front_stream(p) ...
 *                       scalce_pipeline_submit(p, slot, flush_now, &launched);
 *     scalce_pipeline_drain(p);          (the caller walks the slots for their outputs, or retires them one by one before)
 * flush_now: 0 = the launch goes out when `group` shards are pending; 1 = what is pending goes out now and nothing will run
 * beside it (the end of a run, or a wave that fills every slot -- `group` may be as large as nslots for a caller that plans
 * its launches): it is shaped for its own latency, on every CU; 2 = it goes out now, front stages of further shards follow
 * beside it.  A coder launch takes ~0.5 s whether it holds five 50 M-read shards or fifteen: a caller that knows the length
 * of its run puts the remainder first and then launches whole waves (bench.py --launch-plan waves). */
typedef struct scalce_pipeline scalce_pipeline;
int scalce_pipeline_create(scalce_batch **batches, int nslots, int group, int coder_streams, int external_coder, scalce_pipeline **out);
void scalce_pipeline_destroy(scalce_pipeline *p);
const char *scalce_pipeline_error(const scalce_pipeline *p);
void *scalce_pipeline_front_stream(scalce_pipeline *p);
void *scalce_pipeline_coder_stream(scalce_pipeline *p, int i);
int scalce_pipeline_acquire(scalce_pipeline *p, int *slot, int *retired);
int scalce_pipeline_submit(scalce_pipeline *p, int slot, int flush_now, int *launched);
int scalce_pipeline_retire(scalce_pipeline *p, int slot, int *had_shard);
int scalce_pipeline_flush_last(scalce_pipeline *p);  /* what is pending goes out as the last launch of a run */
int scalce_pipeline_drain(scalce_pipeline *p);

#ifdef __cplusplus
}
#endif
#endif /* SCALCE_HIP_H */
