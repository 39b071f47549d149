/*
 * msr.h — C-ABI of the MI355X-native learned-sparse retrieval scorer.
 *
 * This is the drop-in boundary for the reference's sparse search step. In the reference
 * (cjc20000323/mllm_sparse_retrieval) that step is three calls into pyserini/Anserini/Lucene:
 *
 *     LuceneImpactSearcher(index_dir, None)              src/search.py:273
 *     searcher.set_analyzer(JWhiteSpaceAnalyzer())       src/search.py:274-275
 *     searcher.batch_search(queries, qids, k, threads)   src/search.py:86-87
 *
 * plus the offline index build `python -m pyserini.index.lucene --impact --pretokenized`
 * (scripts/sparse_index.sh:12-18) over the jsonl written by src/encode.py:351-359,426.
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 (MSR_OK) or a negative MSR_E_* code and
 *     never throws; msr_last_error() returns a thread-local message for the last failure.
 *   - the caller allocates every output array; the library owns only opaque handles and device buffers.
 *   - one in-flight call per handle (each index handle owns one HIP stream); handles are not thread-safe.
 *   - all integers little-endian, fixed width. Doc "ordinals" are ranks of the external doc-id strings in
 *     bytewise ascending order, so "lower ordinal wins a score tie" is the declared tie rule T1
 *      by construction.
 *   - scoring is exact unsigned 32-bit integer arithmetic on the GPU; a query whose worst-case score
 *     could exceed 2^32-1 is rejected with MSR_E_OVERFLOW rather than computed inexactly.
 *   - there is NO CPU scoring path in this library: searching a handle opened with device < 0, or on a
 *     machine without a usable HIP device, fails with MSR_E_NODEVICE.
 */
#ifndef MSR_H
#define MSR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSR_OK 0
#define MSR_E_INVAL (-1)    /* bad argument */
#define MSR_E_IO (-2)       /* file could not be read / written */
#define MSR_E_FORMAT (-3)   /* malformed jsonl line or index file */
#define MSR_E_NOMEM (-4)    /* host or device allocation failed */
#define MSR_E_NODEVICE (-5) /* no HIP device bound to this handle */
#define MSR_E_HIP (-6)      /* a HIP runtime call failed */
#define MSR_E_OVERFLOW (-7) /* worst-case score of a query exceeds 2^32-1 */
#define MSR_E_RANGE (-8)    /* term id / weight / k outside the supported range */
#define MSR_E_COMM (-9)     /* RCCL failure */

/* search flags */
#define MSR_F_DROP_DF_EQ_N 1u /* drop query terms present in every doc (pyserini idf>min_idf filter, contract T3) */

#define MSR_KMAX 1024 /* largest supported k (reference default depth is 1000, src/arguments.py:59) */

typedef struct msr_index msr_index; /* an opened (optionally device-resident) inverted index */
typedef struct msr_batch msr_batch; /* a device-resident batch of CSR queries + its result buffers */

typedef struct msr_info {
    uint64_t n_docs;        /* docs in the whole index */
    uint64_t n_postings;    /* (term,doc) pairs with weight > 0 in the whole index */
    uint64_t n_vecs;        /* 16-byte posting vectors stored (postings + per-segment padding) */
    uint32_t n_terms;       /* dictionary size */
    uint32_t tile_docs;     /* docs per tile (accumulator tile held in LDS) */
    uint32_t n_tiles;       /* tiles in the whole index */
    uint32_t max_weight;    /* largest stored weight */
    uint32_t shard_tile0;   /* first tile resident on this handle */
    uint32_t shard_ntiles;  /* tiles resident on this handle */
    int32_t device;         /* HIP device ordinal, or -1 */
    uint32_t n_dense;       /* terms stored in the dense head */
    uint32_t term_lo;       /* term-range shards: terms [term_lo, term_hi) are resident (else 0 .. n_terms) */
    uint32_t term_hi;
    uint64_t resident_bytes; /* bytes of the index (segment table + postings + dense head) this handle holds in HBM */
} msr_info;

/* ---- index build: replaces scripts/sparse_index.sh:12-18 (pyserini.index.lucene --impact --pretokenized) ----
 * Reads every *.jsonl / *.json file of `jsonl_dir` (lines {"id":…,"content":…,"vector":{tok:int}} as written by
 * src/encode.py:351-359,426) and writes one index file. tile_docs = 0 picks the default (8192); supported: 4096, 8192, 12288, 16384, 32768. */
int msr_index_build(const char* jsonl_dir, const char* out_path, int threads, uint32_t tile_docs);

/* Process-wide build options (set before building; not thread-safe):
 *   "dense_min_density" (default 0.4): a term with df >= density * n_docs is stored in the doc-major dense head
 *   "dense_max_terms"   (default 16, at most 32; 0 disables the dense head)
 *   "tie_order"         (default 0: doc ordinals = rank of the doc-id string, so a score tie goes to the bytewise lower
 *                        doc id, contract T1; 1: ordinals = input order, a tie goes to the doc indexed first) */
int msr_set_build_option(const char* key, double value);

/* Same index from a doc-major CSR already in memory (used by the synthetic encode step and the benchmark).
 * doc_ids / term_strs may be NULL: docs are then named by their decimal row number and terms by their decimal id. */
int msr_index_build_csr(const char* out_path, uint64_t n_docs, uint32_t n_terms, const uint64_t* doc_ptr,
                        const uint32_t* term_id, const uint32_t* weight, const char* const* doc_ids,
                        const char* const* term_strs, int threads, uint32_t tile_docs);

/* ---- open / close: replaces LuceneImpactSearcher(index_dir, None), src/search.py:273 ----
 * device >= 0 uploads the postings to that HIP device; device = -1 opens host-side metadata only
 * (dictionary, df, doc ids) — such a handle cannot search. */
int msr_index_open(const char* path, int device, msr_index** out);

/* Doc-range shard `shard` of `n_shards` (contiguous tile range): only that slice of the
 * postings is uploaded. Ordinals in results stay global. */
int msr_index_open_shard(const char* path, int device, int shard, int n_shards, msr_index** out);

/* Term-range shard `shard` of `n_shards` (the north star's partition, BASELINE.json configs[3]; no counterpart in the
 * reference, which opens one whole index per rank, src/search.py:216,273): term ranges are contiguous in term id and
 * balanced by postings; of every doc tile only the segments of the owned terms, the matching columns of the segment
 * table and the dense-head pairs that hold an owned term are uploaded. Such a handle serves the term-range protocol only
 * (msr_batch_create_termshard with the same shard / n_shards, msr_batch_search_termshard,
 * msr_search_termshard_emulated_handles); a plain msr_batch_create on it is refused. */
int msr_index_open_termshard(const char* path, int device, int shard, int n_shards, msr_index** out);

/* Closing an index releases the device buffers of every batch still alive on it and detaches those batches: they
 * fail with MSR_E_INVAL afterwards and msr_batch_destroy only frees the host object. */
void msr_index_close(msr_index* ix);
int msr_index_info(const msr_index* ix, msr_info* info);

/* token -> term id (-1 = not in the index vocabulary); the host-side half of pyserini's query encoding
 * that src/search.py:86-87 triggers. */
int msr_term_lookup(const msr_index* ix, const char* const* toks, int n, int32_t* term_ids);
int msr_term_df(const msr_index* ix, const int32_t* term_ids, int n, uint32_t* df);
int msr_term_str(const msr_index* ix, uint32_t term_id, const char** s);
/* external doc id of an ordinal: `hit.docid`, src/search.py:97. Borrowed pointer, valid until close. */
int msr_docid_str(const msr_index* ix, uint32_t ord, const char** s);

/* ---- search: replaces searcher.batch_search(queries, qids, k, threads), src/search.py:86-87 ----
 * Queries are CSR: query i owns entries [q_ptr[i], q_ptr[i+1]) of (q_term, q_w). Entries with q_term < 0
 * (out-of-vocabulary) or q_w <= 0 are ignored; duplicate terms add (src/search.py:419-422).
 * Outputs (row-major [nq][k]): doc ordinals, scores as f32 (`hit.score`), optional exact u32 scores,
 * and out_n[i] = number of hits of query i (only docs with score > 0 are hits, so out_n[i] <= k). */
int msr_search_csr(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq, int k,
                   uint32_t flags, uint32_t* out_doc_ord, float* out_score, uint32_t* out_score_u32,
                   int32_t* out_n);

/* Where the calling thread's last msr_search_csr spent its time, in microseconds: {query normalisation + device buffers
 * + upload enqueued, kernels enqueued, wait for the stream, download + copy out, release, the whole call, HIP-event span
 * of the scoring kernel(s), of the merge}. The reference calls batch_search with 4 queries at a time
 * (scripts/search_sparse.sh:16): such a call is launch- and copy-latency, not kernel time. */
int msr_search_laps(double out_us[8]);

/* The same search for query STRINGS as the reference builds them (each token repeated `weight` times,
 * src/search.py:419-422): whitespace split, token counts and dictionary lookup happen on the host inside this call —
 * the work pyserini's batch_search does in Python before handing the query to Lucene. */
int msr_search_text(msr_index* ix, const char* const* queries, int nq, int k, uint32_t flags, uint32_t* out_doc_ord,
                    float* out_score, uint32_t* out_score_u32, int32_t* out_n);

/* The host half of that call on its own: query strings -> CSR of (term id, count), out-of-vocabulary tokens dropped
 * (what pyserini's query encoding does in Python before Lucene sees the query, behind src/search.py:86-87); for callers
 * that hand the CSR to msr_hybrid_search / msr_batch_create. q_ptr (nq + 1 entries) is always filled and *n_entries is
 * set to the number of (term, count) pairs; q_term / q_w are filled when cap >= *n_entries (call with cap = 0 to size). */
int msr_encode_queries(const msr_index* ix, const char* const* queries, int nq, int64_t* q_ptr, int32_t* q_term,
                       int32_t* q_w, int64_t cap, int64_t* n_entries);

/* ---- resident batches: the same search with inputs and outputs kept in HBM (benchmark, pipelining) ---- */
int msr_batch_create(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq,
                     int kmax, uint32_t flags, msr_batch** out);
int msr_batch_search(msr_batch* b, int k); /* enqueue on the index's stream; returns before completion */
int msr_batch_sync(msr_batch* b);
int msr_batch_fetch(msr_batch* b, uint32_t* out_doc_ord, float* out_score, uint32_t* out_score_u32,
                    int32_t* out_n); /* syncs, then copies the last search's results ([nq][k]) */
/* HIP-event durations of the last msr_batch_search on its own stream (ms): the scoring kernel and the merge. */
int msr_batch_kernel_ms(msr_batch* b, float* score_ms, float* merge_ms);
/* Sums over every msr_batch_search call since the last reset (each call records its own HIP events on the
 * index's stream), so a whole timed region can be priced without a host sync per call. */
int msr_batch_timing_reset(msr_batch* b);
int msr_batch_timing_sum(msr_batch* b, int* n_calls, float* score_ms, float* merge_ms);
/* Diagnostic builds only (env MSR_DEBUG_FLAGS bit 3): summed s_memtime deltas of wave 0 per kernel phase
 * {zero, stage, stream, wait, maxima, candidates, rank, -}. All zeros when the diagnostic is off. */
int msr_batch_debug_stamps(msr_batch* b, unsigned long long out[8]);
/* Algorithmic bytes of one search of this batch, SURVEY.md §8d:
 * sum over queries of  sum_t df(t)*(4+2) + |q|*12 + k*8  (df restricted to this handle's shard). */
int msr_batch_algo_bytes(const msr_batch* b, int k, uint64_t* bytes, uint64_t* postings);
/* The work one search of this batch does, for the roofline of the scoring kernel (no reference counterpart):
 * out = {postings walked through inverted lists (one LDS add each), postings scored out of the dense head (half a
 * v_dot2_u32_u16 each), (tile, query) workgroups, LDS bytes written to initialise the accumulator tiles, LDS bytes the
 * selection reads back (two passes), kept query entries}. */
int msr_batch_work(const msr_batch* b, uint64_t out[6]);
void msr_batch_destroy(msr_batch* b);

/* ---- multi-GPU exchange (doc-range shards, one RCCL all-gather of per-shard top-k) ----
 * No counterpart in the reference (it never shards the sparse index, src/search.py:216,273). */
#define MSR_COMM_ID_BYTES 128
int msr_comm_unique_id(char id[MSR_COMM_ID_BYTES]);
int msr_comm_init(msr_index* ix, int n_ranks, int rank, const char id[MSR_COMM_ID_BYTES]);
/* local search + all-gather + merge; afterwards msr_batch_fetch returns the GLOBAL top-k on every rank. */
int msr_batch_search_sharded(msr_batch* b, int k);
int msr_comm_destroy(msr_index* ix);
/* What the communicator of `ix` reports about itself (ncclCommCount / ncclCommUserRank / ncclCommCuDevice). */
int msr_comm_info(const msr_index* ix, int* n_ranks, int* rank, int* device);
/* One line of text for logs and the benchmark record: HIP runtime version, RCCL version and the path of the librccl
 * the process actually resolved (dladdr). */
int msr_runtime_info(char* buf, int cap);
/* hipDeviceSynchronize on `device` (the benchmark's device-wide fence, independent of any torch state). */
int msr_device_sync(int device);
/* Bandwidth of a device-to-device copy of `bytes` (read + write bytes / time, GB/s): the measured HBM figure quoted
 * beside the vendor peak in the benchmark record. */
int msr_device_copy_gbs(int device, uint64_t bytes, int reps, double* gbs);
/* Measured peaks of the pipes the scoring kernel's useful work runs on, at its launch shape (512 threads, four
 * workgroups per CU): out = {ds_add_u32 lane operations / s, v_dot2_u32_u16 lane operations / s, LDS bytes / s at one
 * 16-byte write per two 16-byte reads, CUs}. Takes a few milliseconds. */
int msr_device_peak_rates(int device, double out[4]);

/* Term-range shards (the north star's partition; exact protocol of DESIGN.md §6): the batch holds only the query terms
 * of term range `shard` of `n_shards` (ranges are contiguous in term id and balanced by postings). Search = dump the
 * partial accumulators, ncclReduceScatter(sum) them into doc ranges, select, all-gather, merge. The handle must hold
 * every doc tile (msr_index_open) and the communicator's rank / size must equal shard / n_shards. */
int msr_batch_create_termshard(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq,
                               int kmax, uint32_t flags, int shard, int n_shards, msr_batch** out);
int msr_batch_search_termshard(msr_batch* b, int k);
/* The same protocol for `n_shards` logical term shards played on one GPU (sums in place instead of RCCL). */
int msr_search_termshard_emulated(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq,
                                  int k, uint32_t flags, int n_shards, uint32_t* out_doc_ord, float* out_score,
                                  uint32_t* out_score_u32, int32_t* out_n);
/* ... and with one handle per logical shard, shards[g] opened by msr_index_open_termshard(path, dev, g, n_shards) on
 * the same device: every shard scores out of its own partial residency. */
int msr_search_termshard_emulated_handles(msr_index* const* shards, int n_shards, const int64_t* q_ptr,
                                          const int32_t* q_term, const int32_t* q_w, int nq, int k, uint32_t flags,
                                          uint32_t* out_doc_ord, float* out_score, uint32_t* out_score_u32, int32_t* out_n);

/* Merge `n_lists` per-shard result lists (each [nq][k] as written by msr_batch_fetch) on the device of `ix`
 * with the same tie rule; used by the host-side exchange (torch.distributed all_gather) and by tests. */
int msr_merge_lists(msr_index* ix, int n_lists, int nq, int k, const uint32_t* doc_ord, const uint32_t* score_u32,
                    const int32_t* n, uint32_t* out_doc_ord, float* out_score, uint32_t* out_score_u32,
                    int32_t* out_n);

/* ---- dense flat inner-product search (hybrid path): replaces tevatron FaissFlatSearcher / faiss IndexFlatIP with
 * fp16 storage (src/search.py:232-237,254-270; call site search_queries src/search.py:55-63).
 * p_fp16 / q_fp16 are row-major IEEE fp16 matrices ([n][h] / [nq][h], h a multiple of 32); scores are f32-accumulated
 * on MFMA. Outputs [nq][k]: row indices (0xFFFFFFFF padding), order-preserving u32 keys of the f32 scores
 * (key = bits ^ 0x80000000 for non-negative, ~bits for negative floats; 0 = padding) and the hit count. */
typedef struct msr_dense msr_dense;
int msr_dense_open(const uint16_t* p_fp16, uint64_t n, uint32_t h, int device, msr_dense** out);
int msr_dense_search(msr_dense* dx, const uint16_t* q_fp16, int nq, int k, uint32_t* out_idx, uint32_t* out_key,
                     int32_t* out_n, float* gemm_ms, float* select_ms);
void msr_dense_close(msr_dense* dx);
/* Bookkeeping of a dense handle: out = {hipMalloc calls made for its scratch so far, scratch bytes it holds, rows, dim}.
 * The scratch of msr_dense_search / msr_hybrid_search stays with the handle and is only ever grown, so repeated calls of
 * one shape allocate nothing (the reference searches in batches of 2, scripts/search.sh:29). */
int msr_dense_stats(const msr_dense* dx, uint64_t out[4]);
/* Host helper for the two calls above: n f32 values -> IEEE fp16, round to nearest even (what numpy's astype(float16)
 * and the reference's .half() produce: src/search.py:257; bit-identical for every non-NaN input — a NaN comes out
 * quiet, numpy keeps its payload), on `threads` host threads (<= 0: all). The reference hands
 * its query matrix over in f32 (src/search.py:342-343); converting 25 010 x 4 096 values in numpy takes longer than
 * the whole GPU search. */
int msr_f32_to_f16(const float* src, uint16_t* dst, uint64_t n, int threads);

/* ---- hybrid search on the GPU: sparse top-`depth` + dense top-`depth` + the reference's min-max fusion
 * (fuse, src/hybrid.py:32-53, weights [alpha, 1-alpha] src/search.py:459) + top-k, without leaving HBM in between.
 * row2ord[r] = sparse doc ordinal of dense row r; self_ord[q] (nullable) = ordinal removed from query q's lists
 * (remove_query, src/search.py:72-74) or -1. row2ord must be a permutation of the ordinals. depth, k <= MSR_KMAX.
 * The GEMM runs on the passage rows permuted into ordinal order (built once per mapping, kept on the dense handle), so a
 * query's row of dense scores lines up with the sparse accumulator tiles, and no list ever leaves HBM:
 *   - one tile (n_docs <= 8192), k <= 64: ONE kernel per query does sparse scoring, both depth selections, fusion and
 *     top-k (hybrid_tiles); ms = {that kernel, dense GEMM, 0, 0};
 *   - otherwise (the reference's own hybrid run: 25 010 caption docs = 4 tiles, scripts/search.sh): one workgroup per
 *     (tile, query) scores the tile and emits its quota of candidates for both depth lists, one workgroup per query
 *     finds both depth-th bests among them, fuses and ranks; a query whose depth list a tile's quota did not cover
 *     (verified per query) is repeated with every tile's own top-depth, so the result is exact whatever the quota;
 *     ms = {candidate kernel, dense GEMM, 0, fusion kernel};
 *   - tile sizes other than 4096 / 8192 docs, shard handles, MSR_NO_FUSED_HYBRID: the list-based path (score_tiles at
 *     k = depth, select_tiles over the score matrix, fuse_tiles, merges); ms = {sparse, dense GEMM, dense select, fusion}.
 * Tie rule of the DENSE depth list (two passages with bit-equal scores at the depth boundary — duplicated captions do
 * that): the first two paths keep the lower doc ORDINAL, the list-based path and msr_dense_search the lower ROW, as a
 * flat index does. Everything else (sparse ties, fused-score ties: lower ordinal) is the same on every path. */
int msr_hybrid_search(msr_index* ix, msr_dense* dx, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w,
                      const uint16_t* q_fp16, int nq, int depth, int k, float alpha, uint32_t flags,
                      const uint32_t* row2ord, const int32_t* self_ord, uint32_t* out_ord, float* out_score, int32_t* out_n,
                      float ms[4]);

/* ---- encode-side sparsifier: log(1 + relu(logit)) -> top-k -> rint(x * 100) (src/model.py:104, src/encode.py:69-75).
 * logits: row-major [rows][vocab], f32 (is_f16 = 0) or IEEE fp16 (is_f16 = 1). fp16_math = 1 rounds 1 + relu and the
 * log to half like a model running in fp16. Outputs [rows][k]: vocabulary ids (ties: lower id first), the selected
 * values v and the integer weights rint(v * 100). */
int msr_sparsify(const void* logits, int is_f16, int fp16_math, int rows, uint32_t vocab, int k, int device,
                 uint32_t* out_idx, float* out_val, int32_t* out_weight);

/* ---- synthetic encode step (SURVEY.md §8d generator; stands in for src/encode.py when no MLLM is present) ----
 * Fills a doc-major CSR of n vectors with `nnz` distinct terms each, drawn without replacement from
 * p(r) ~ r^-zipf_s over n_terms, weights max(1, rint(100*ln(1+x))), x ~ LogNormal(0.5, 0.6), clipped to [1,400].
 * ptr has n+1 entries; term/weight have n*nnz. Deterministic in (seed, row) regardless of threads. */
int msr_synth_vectors(uint64_t n, uint32_t nnz, uint32_t n_terms, double zipf_s, uint64_t seed, int threads,
                      uint64_t* ptr, uint32_t* term, uint32_t* weight);

const char* msr_last_error(void);
const char* msr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MSR_H */
