/*
 * dawn_hip.h — C ABI of libdawn_hip.so: the MI355X (gfx950) drop-in for DawnSearch's
 * embed-and-rank hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * dawn-search/dawnsearch repository).  The reference reaches its vector index through
 * `usearch::ffi` (cxx bridge) and its embedder through `EmbeddingProvider`; a Rust `extern "C"`
 * block binding exactly these symbols is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns DAWN_OK (0) or a negative DAWN_ERR_* code; dawn_last_error() returns a
 *     thread-local message for the last failing call on this thread (the reference surfaces
 *     `cxx::Exception` / `anyhow::Error` text the same way);
 *   - the caller owns every host buffer; the library owns all device memory;
 *   - a handle is used from one thread at a time (the reference drives each provider from one
 *     dedicated blocking thread: src/bin/dawnsearch.rs:63-66,76-78); distinct handles are independent;
 *   - there is NO CPU fallback: without a usable HIP device every create call fails with
 *     DAWN_ERR_NO_DEVICE.
 */
#ifndef DAWN_HIP_H
#define DAWN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: only what these headers declare is exported */
#endif

#define DAWN_OK 0
#define DAWN_ERR_INVALID_ARG (-1)
#define DAWN_ERR_NOT_NORMALIZED (-2) /* "Search vector is not normalized" search_provider.rs:206-208,265-267 */
#define DAWN_ERR_HIP (-3)
#define DAWN_ERR_IO (-4)
#define DAWN_ERR_NO_DEVICE (-5)
#define DAWN_ERR_UNSUPPORTED (-6)
#define DAWN_ERR_OOM (-7)

#define DAWN_EM_LEN 384 /* src/search/vector.rs:26 */
#define DAWN_MAX_K 64   /* largest `count` of one search call (reference uses 20: search_provider.rs:214) */

#define DAWN_DTYPE_F32 0  /* ScalarKind::F32, search_provider.rs:38 */
#define DAWN_DTYPE_BF16 1 /* rows stored as bf16 (768 B): twice the rows per GB of HBM (1B x 384 on 8 GPUs).  Vectors
                           * are given and gated as f32, rounded to nearest-even on add; distances are
                           * 1 - sum(q_i * bf16(x_i)) in the same sequential f32 arithmetic, exact for the stored rows */

const char *dawn_last_error(void);
int dawn_version(void);
int dawn_device_count(int *count);

/* ------------------------------------------------------------------------------------------ */
/* Vector index — replaces usearch::ffi::Index as used by src/search/search_provider.rs          */
/* ------------------------------------------------------------------------------------------ */
typedef struct dawn_index dawn_index;

/* new_index(&INDEX_OPTIONS) — search_provider.rs:35-42,102.  dims must be 384, metric is IP
 * (distance = 1 - sum(q_i*x_i)), dtype DAWN_DTYPE_F32 or DAWN_DTYPE_BF16.  `device` = HIP device ordinal. */
int dawn_index_create(size_t dims, int dtype, int device, dawn_index **out);
/* The same index with its rows dealt over n_gpus devices of one node (dawn_index_create(dim, dtype, n_gpus)):
 * devices[i] = HIP ordinal of shard i (NULL = 0..n_gpus-1; devices[0] is the root, where queries arrive and results
 * leave).  EVERY dawn_index_* call below takes the handle unchanged: add / add_batch deal chunks of 4096 consecutive rows
 * round robin, search runs all shards concurrently, gathers the per-shard top-k blobs (one grouped ncclAllGather over
 * RCCL/xGMI; peer copies when "shard_gather" = 2 or a device holds several shards) and merges them on the root — ties go to
 * the earlier-added row, so the answer is bit-identical to the single-device index; save / load files are the same
 * files.  This is the multi-GPU counterpart of search_remote's fan-out + BestResults merge (search_service.rs:201-277)
 * for ONE process driving the GPUs of a node (the one-process-per-GPU form is dawn_index_search_device +
 * dawn_topk_merge_packed_device around the caller's own collective).  Extra options: "shard_chunk" (while empty),
 * "shard_gather" 0 auto / 1 RCCL / 2 peer copies, "shard_threads" 1 / 0: one issuing host thread per shard beyond the first
 * (default when every shard has its own device) or the caller's thread alone. */
int dawn_index_create_sharded(size_t dims, int dtype, int n_gpus, const int *devices, dawn_index **out);
/* *n_shards (1 for a plain index); *gather: 1 RCCL all-gather in use, -1 RCCL selected and not initialised yet (first
 * search), 2 peer copies, 0 nothing to gather; shard_sizes[min(n_shards, cap)] rows per shard.  NULL = not wanted. */
int dawn_index_shard_info(dawn_index *idx, int *n_shards, int *gather, size_t *shard_sizes, size_t cap);
void dawn_index_destroy(dawn_index *idx); /* drop of UniquePtr<Index> */

int dawn_index_reserve(dawn_index *idx, size_t capacity);      /* index.reserve(n)  :133,282 */
size_t dawn_index_size(const dawn_index *idx);                 /* index.size()      :246,280 */
size_t dawn_index_capacity(const dawn_index *idx);             /* index.capacity()  :280     */

/* index.add(id, &q) :149,284.  v = 384 f32, must pass is_normalized (vector.rs:185-192) — the
 * reference checks this right before every add (:147 bytes_to_embedding, :265-267). Grows like
 * usearch after an explicit reserve; also grows on its own when full.  The row is gated on the host and STAGED in pinned
 * host memory; up to 1024 staged rows travel to the GPU together (one transfer, one shadow update, one synchronisation)
 * when the stage is full or in front of the next call that looks at the rows (search, save, get_rows, add_batch, ...):
 * the reference's one-row-per-call insert and rebuild loops (:127-153, :280-284) cost well under a microsecond per row.
 * dawn_index_size counts staged rows; a growth failure (out of HBM) is reported by the call that flushes. */
int dawn_index_add(dawn_index *idx, uint64_t id, const float *v);
/* Bulk form of fill_index_from_db's loop (:135-150): n rows in one transfer + one validation kernel.
 * On a non-normalised row nothing is added and DAWN_ERR_NOT_NORMALIZED is returned. */
int dawn_index_add_batch(dawn_index *idx, size_t n, const uint64_t *ids, const float *v);

/* index.search(query, count) -> Matches{labels, distances} :214.  Exact: distances[i] =
 * 1.0f - (sequential f32 sum of q_i*x_i) bit-for-bit as vector.rs:128-134, ascending, ties ->
 * earlier-added row.  *found = min(count, size).  count <= DAWN_MAX_K.  The query must pass
 * is_normalized (:206-208). */
int dawn_index_search(dawn_index *idx, const float *query, size_t count, uint64_t *labels,
                      float *distances, size_t *found);
/* The answering side of a remote search (src/net/udp_service.rs:174-215): the same search, then only the hits with
 * distance < distance_limit are reported ("if page.distance >= d { continue }", :196-199) — *found counts them; the
 * limit a peer sends is its BestResults::worst_distance(), 0.0 until it holds 20 local results (best_results.rs:40). */
int dawn_index_search_limited(dawn_index *idx, const float *query, size_t count, float distance_limit,
                              uint64_t *labels, float *distances, size_t *found);
/* B queries in one call (the reference has no batching; this is what a batching caller binds).
 * queries [B][384]; labels/distances [B][count]; found [B]. */
int dawn_index_search_batch(dawn_index *idx, const float *queries, size_t B, size_t count,
                            uint64_t *labels, float *distances, size_t *found);

/* index.save(path) :117,178.  Atomic: written to `path`.tmp, fsync'ed, renamed — an interrupted save leaves the old file. */
int dawn_index_save(dawn_index *idx, const char *path);
/* index.load(path) :115.  Replaces the contents; all or nothing: ANY failure — a missing file, a bad header, a truncated /
 * corrupt file, a row failing the is_normalized gate — leaves the index EMPTY (never partially filled, never its old
 * rows), so the reference's
 * `if !load(path).is_ok() { fill_index_from_db() }` (:115-117) rebuilds onto a clean index.  The file streams through
 * pinned host staging, reads overlapped with the DMA into HBM. */
int dawn_index_load(dawn_index *idx, const char *path);
/* Bulk-load the packed PageEntry file of src/index/warc.rs:35-43 (1568-B records, vector at
 * byte 16) as read by examples_old/document_embeddings.rs:56-71; ids = first_id + record index.  Appends; all or
 * nothing.  Records go to the GPU as they are on disk and are cut down to their vectors there. */
int dawn_index_load_page_entries(dawn_index *idx, const char *emb_path, uint64_t first_id);

/* ---- device-resident forms (queries/results already in HBM; nothing is synchronised) -------- */
/* d_queries [B][384] f32, d_labels [B][count] u64, d_distances [B][count] f32, d_found [B] u32 are
 * DEVICE pointers on the index's (root) device; `stream` is a hipStream_t (NULL = default stream).
 * Queries are assumed validated.  For B <= 256 on a single-device index this is kernel launches only — no allocation, no
 * synchronisation — and every rung of the ladder behind a failed certificate is a launch PREDICATED on the query's flag on the
 * device: the sequence can be captured into a hipGraph and replayed, and a replay is exact for any query, whatever rung it needs.
 * What is decided on the HOST, per call, is only which of two exact sequences is issued — the filter stream first, or (an index
 * whose certificates fail often: the ladder feedback, read from counters the device mirrors into pinned memory; never a
 * synchronisation) the bounded pass directly, and how deep the batched pass aims.  A captured graph therefore FREEZES the sequence
 * chosen at capture: it stays exact, but it no longer adapts — capture with option "ladder_feedback" = 0 (always the filter stream
 * first) or = 2 (always the bounded pass directly) to choose it explicitly, or re-capture now and then.
 * (Workspaces and the filter shadows of the current rows are prepared by create / add / load / reserve / set_option, which
 * synchronise; a caller searching on its own stream must have that stream idle before it mutates the index.  B > 256 in one call
 * grows the workspaces once.) */
int dawn_index_search_device(dawn_index *idx, const float *d_queries, size_t B, size_t count,
                             uint64_t *d_labels, float *d_distances, uint32_t *d_found, void *stream);
/* Stable G-way merge of per-shard results (each ascending by (distance, shard-local order)) —
 * the multi-GPU counterpart of search_service.rs:214-263 (BestResults merge of local + remote).
 * d_in_labels/d_in_distances [G][B][count] (e.g. the all-gather output), d_in_found [G][B];
 * outputs [B][count] / [B].  Ties -> lower shard, then shard-local order. */
int dawn_topk_merge_device(int device, size_t G, size_t B, size_t count, const uint64_t *d_in_labels,
                           const float *d_in_distances, const uint32_t *d_in_found, uint64_t *d_labels,
                           float *d_distances, uint32_t *d_found, void *stream);

/* One-collective form.  Each shard writes its results into ONE blob of dawn_result_blob_bytes(B, count) bytes:
 *   labels u64 [B][count] at +0 | distances f32 [B][count] at +B*count*8 | found u32 [B] at +B*count*12
 * (dawn_index_search_device accepts pointers into such a blob), the ranks all-gather the blobs (one
 * ncclAllGather over xGMI), and d_blobs = the G gathered blobs back to back. */
size_t dawn_result_blob_bytes(size_t B, size_t count);
int dawn_topk_merge_packed_device(int device, size_t G, size_t B, size_t count, const void *d_blobs,
                                  uint64_t *d_labels, float *d_distances, uint32_t *d_found, void *stream);

/* Host form of the same stable merge (host pointers; used where the lists already sit in host memory,
 * e.g. the reference's own local+remote merge point, and by the CPU/gloo tests of the sharded path). */
int dawn_topk_merge_host(size_t G, size_t B, size_t count, const uint64_t *in_labels, const float *in_distances,
                         const uint32_t *in_found, uint64_t *labels, float *distances, uint32_t *found);

/* Counters: searches that needed the exact fallback pass (certificate failed).  Counted on the device at the end of every
 * search, whichever entry point issued it (host API or dawn_index_search_device); reading synchronises the device. */
int dawn_index_stats(dawn_index *idx, uint64_t *searches, uint64_t *fallbacks);
/* ... and searches whose 64-row certificate failed but a deeper one held (no exact pass): deeper rounds of the same
 * certificate (128 .. 256 rows) or, after those, the 1024-row second chance. */
int dawn_index_stats_ext(dawn_index *idx, uint64_t *searches, uint64_t *second_chances, uint64_t *fallbacks);
/* ... of the second_chances, those a deeper round settled (the cheap kind: ~10 us per round). */
int dawn_index_stats_deep(dawn_index *idx, uint64_t *deepened);
/* ... and the ladder behind the certificates.  bounded: queries whose certificates all failed and which the BOUNDED EXACT PASS
 * answered — one stream of the int8 shadow plus an exact score for every row whose upper bound reaches the k-th best distance
 * found so far: no lists, cannot fail, costs the int8 stream + the rows that crowd the top of the query (topical data: the dense
 * part of a cluster); not counted in `fallbacks` (the exact pass over all rows did not run for them).  packed_failures: single
 * queries whose packed-stream certificate failed.  demoted: single queries the index sent to the bounded pass directly because
 * the packed certificate had been failing for more than a third of the recent ones (ladder feedback; they count as bounded too).
 * Any pointer may be NULL. */
int dawn_index_stats_ladder(dawn_index *idx, uint64_t *bounded, uint64_t *packed_failures, uint64_t *demoted);
/* HBM held by the index, in bytes: its rows (reserve()'d capacity; usearch: memory_usage()), the filter shadows built
 * so far (int8: 384 B/row + 8 B per 32 rows; packed 5- / 6-bit: 240 / 288 B/row + 8 B per 32 rows, indexes of >= 768 Ki rows; f16: 768 B/row),
 * everything else (labels, search workspaces, staging). */
int dawn_index_memory(dawn_index *idx, uint64_t *rows_bytes, uint64_t *shadow_bytes, uint64_t *other_bytes);
/* Per-index options (name, value).  A caller that binds this header needs none of them: the defaults are the tuned values and
 * results never depend on an option.  The catalogue (tuning knobs the tools sweep, A/B switches of the ladder behind the
 * certificates, test hooks) is in dawn_hip_debug.h. */
int dawn_index_set_option(dawn_index *idx, const char *name, int64_t value);

/* ------------------------------------------------------------------------------------------ */
/* src/search/vector.rs + best_results.rs host helpers (bit-compatible with the reference)      */
/* ------------------------------------------------------------------------------------------ */
int dawn_vec_is_normalized(const float *v /*[384]*/);       /* vector.rs:185-192 -> 1/0 */
/* the same predicate over n vectors [n][384]: index of the first that fails it, or n (what dawn_index_search_batch runs on its queries) */
size_t dawn_vec_first_not_normalized(const float *v, size_t n);
void dawn_vec_normalize(float *v, size_t n);                /* vector.rs:194-197 */
void dawn_vec_to24(const float *v, uint8_t *out /*[1152]*/);/* vector.rs:74-86  */
int dawn_vec_from24(const uint8_t *in, float *out);         /* vector.rs:57-72; DAWN_ERR_NOT_NORMALIZED */

typedef struct dawn_best_results dawn_best_results;          /* best_results.rs:28-33 */
int dawn_best_new(size_t size, dawn_best_results **out);     /* :36-43 */
void dawn_best_free(dawn_best_results *b);
int dawn_best_insert(dawn_best_results *b, size_t id, float distance); /* :44-65 -> 1 inserted / 0 not / < 0 error */
void dawn_best_sort(dawn_best_results *b);                   /* :71-79 */
float dawn_best_worst_distance(const dawn_best_results *b);  /* :93-95 (0 until full) */
size_t dawn_best_len(const dawn_best_results *b);            /* :85-87 */
int dawn_best_get(const dawn_best_results *b, size_t i, size_t *id, float *distance);

/* ------------------------------------------------------------------------------------------ */
/* Embedder — replaces EmbeddingProvider (src/embedding/embedding_service.rs:49-139) minus the   */
/* tokenizer: token ids cross the boundary.                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct dawn_embedder dawn_embedder;

/* EmbeddingProvider::new (:55-95): weights from a safetensors file with the tensor names of
 * src/embedding/model.rs:235-255,301-303,359-363,417,443-447 (optional "bert." prefix :543-547,
 * LayerNorm gamma/beta fallback :210-222); config_json as model.rs:115-133 (NULL = all-MiniLM-L6-v2). */
int dawn_embedder_create(const char *safetensors_path, const char *config_json_path, int device,
                         dawn_embedder **out);
void dawn_embedder_destroy(dawn_embedder *e);
/* Host-only check of the files dawn_embedder_create would load (same parsing, same tensor-name resolution, same errors:
 * DAWN_ERR_IO / DAWN_ERR_UNSUPPORTED) — needs no device, so a deployment can validate model.safetensors / config.json
 * before it claims a GPU.  The files are untrusted input: whatever they hold, the call returns a code. */
int dawn_embedder_check_files(const char *safetensors_path, const char *config_json_path);
/* calculate_embedding (:97-139) for B token sequences packed back to back: token_ids[seq_offsets[B]],
 * sequence b = token_ids[seq_offsets[b] .. seq_offsets[b+1]).  Every sequence gets its batch-1
 * result (no padding tokens exist).  out [B][384] unit vectors. */
int dawn_embedder_forward(dawn_embedder *e, const uint32_t *token_ids, const int32_t *seq_offsets,
                          int B, float *out);
/* Device-resident form: d_token_ids/d_seq_offsets/d_out on the embedder's device. total_tokens and
 * max_len are host-known launch geometry. */
int dawn_embedder_forward_device(dawn_embedder *e, const uint32_t *d_token_ids, const int32_t *d_seq_offsets,
                                 int B, int total_tokens, int max_len, float *d_out, void *stream);
/* Per-embedder options; none is needed (catalogue: dawn_hip_debug.h). */
int dawn_embedder_set_option(dawn_embedder *e, const char *name, int64_t value);
/* ------------------------------------------------------------------------------------------ */
/* Host tokenizer — the `tokenizers` crate calls of EmbeddingProvider (embedding_service.rs:88,     */
/* 101-113): Tokenizer::from_file + encode_batch(inputs, add_special_tokens = true).  Pure host   */
/* code (no device needed).  BERT WordPiece pipeline as configured by all-MiniLM-L6-v2's           */
/* tokenizer.json: BertNormalizer (clean, CJK spacing, strip accents, lowercase), BertPreTokenizer,*/
/* WordPiece("##", [UNK], 100 chars/word), [CLS] $A [SEP], truncation to max_length (128).         */
/* ------------------------------------------------------------------------------------------ */
typedef struct dawn_tokenizer dawn_tokenizer;
/* path: a HF tokenizer.json (model.vocab, normalizer.lowercase, truncation.max_length, added_tokens are
 * honoured) or a vocab.txt (one token per line; lowercase, max_length 128). */
int dawn_tokenizer_create(const char *path, dawn_tokenizer **out);
void dawn_tokenizer_destroy(dawn_tokenizer *t);
int dawn_tokenizer_set_max_length(dawn_tokenizer *t, size_t max_length); /* 0 = no truncation */
size_t dawn_tokenizer_vocab_size(const dawn_tokenizer *t);
/* One text -> ids incl. [CLS]/[SEP].  *n = ids needed; DAWN_ERR_INVALID_ARG if cap is too small. */
int dawn_tokenizer_encode(const dawn_tokenizer *t, const char *text_utf8, uint32_t *out_ids, size_t cap, size_t *n);
/* B texts -> packed ids + seq_offsets[B+1]: exactly the inputs of dawn_embedder_forward (no padding: every
 * text keeps its own length, which is what the reference's one-text-per-call path computes). */
int dawn_tokenizer_encode_batch(const dawn_tokenizer *t, const char *const *texts, size_t B, uint32_t *out_ids,
                                size_t cap, int32_t *seq_offsets);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
