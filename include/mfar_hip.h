/*
 * mfar_hip.h -- C ABI of libmfar_hip.so: the MI355X (gfx950) implementation of mFAR's dense multi-field
 * scoring path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * The reference (microsoft/multifield-adaptive-retrieval, pure Python) has no FFI for this path: the seam is
 * ordinary method calls.  Each entry point below names the reference interface it stands behind
 * (file:line relative to the reference root).  The Python classes in
 * multifield-adaptive-retrieval_amd/mfar/ keep the reference's names and signatures and call these via ctypes;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns MFAR_OK (0) or a negative MFAR_ERR_* code and never throws;
 *     mfar_last_error() returns a thread-local description of the last failure on this thread.
 *   - `on_device` != 0: all data pointers of that call are device pointers on the index's device and the work
 *     is enqueued on `stream` (a hipStream_t, may be NULL = default stream) without host synchronisation;
 *     `on_device` == 0: pointers are host pointers, the call copies in/out and synchronises before returning.
 *   - document ids are GLOBAL row numbers (row_offset + local row) = line order of the corpus TSV
 *     (contrastive.py:247-248, modeling/util.py:80-81); they must fit 32 bits (row_offset + n_rows < 2^32-1).
 *   - ordering everywhere: (score descending, doc id ascending)  -- the canonical tie-break; the reference's
 *     own tie order is unspecified (unstable torch.topk, hash-ordered set: contrastive.py:679,696).
 *   - calls on one handle (an index and the pipelines over it) must be serialised by the caller; DIFFERENT handles may be driven from
 *     different host threads at the same time (they share the device's streams inside the library, nothing else).
 */
#ifndef MFAR_HIP_H
#define MFAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFAR_OK 0
#define MFAR_ERR_INVALID (-1)     /* bad argument / shape */
#define MFAR_ERR_HIP (-2)         /* a HIP runtime call failed (no device, launch failure, ...) */
#define MFAR_ERR_NOMEM (-3)       /* device or host allocation failed */
#define MFAR_ERR_UNSUPPORTED (-4) /* valid request this build does not implement */

#define MFAR_DTYPE_F32 0
#define MFAR_DTYPE_BF16 1

#define MFAR_MAX_K 128     /* top-k depth supported by the selection kernels (the reference hard-codes 100) */
#define MFAR_MAX_FIELDS 32 /* prime = 22, amazon = 8, mag = 5 (schema.py:11-53) */

typedef struct mfar_index mfar_index;

/* library / device probes (no reference counterpart).  mfar_version() == MFAR_ABI_VERSION of the header the caller was built
 * against, or the caller must refuse the library: the value changes with every signature change. */
#define MFAR_ABI_VERSION 107
int mfar_version(void);
const char* mfar_last_error(void);
int mfar_device_count(int* n_out);
/* TEST HOOK (process-wide; no reference counterpart): scratch / table allocations of at least `bytes` bytes made by the library fail exactly
 * like a hipMalloc that ran out of device memory (MFAR_ERR_NOMEM, nothing left sticky); 0 switches it off.  Lets the out-of-memory
 * semantics be driven deterministically: with a really full HBM the HIP runtime gives cached resources back when an allocation fails, so
 * whether a given call fits depends on what the process did before (tests/test_gpu_parity.py::test_full_hbm_degrades_to_the_exact_pass uses
 * both).  The slab of mfar_index_create is not affected. */
int mfar_debug_fail_allocations_above(int64_t bytes);

/*
 * The on-HBM sharded index: replaces the per-field float32 np.memmap files `{temp_dir}/{field}.npy`
 * (MemoryMapDict, data/util.py:28-59) and the DenseFlatIndex objects built over them
 * (read_and_create_indices, modeling/util.py:73-108).  One handle holds rows
 * [row_offset, row_offset + n_rows_local) of every field -- the same row split the reference uses for
 * encoding, corpus[n*rank//ws : n*(rank+1)//ws] (contrastive.py:470).
 * dim must be a multiple of 32; n_fields <= MFAR_MAX_FIELDS.  Rows start zero-filled.
 */
int mfar_index_create(mfar_index** out, int device, int64_t n_rows_local, int64_t row_offset, int n_fields, int dim,
                      int dtype);
void mfar_index_destroy(mfar_index* idx);
int mfar_index_info(const mfar_index* idx, int64_t* n_rows_local, int64_t* row_offset, int* n_fields, int* dim,
                    int* dtype, int64_t* slab_bytes);
/* HBM the handle keeps resident between searches, by part: the rows (the slab), the fp16 screen slab of an fp32 index, the 16-bit
 * row-major gather slab (fp32 index: the approximate level of stage 2; bf16 index: the whole-line companion), the unique-row
 * tables / statistics / row norms of the certified stage 1, and the large retained scratch of the pipeline slots: the score dumps of
 * the slots that use one (16-bit codes: rows of the screen slab x 256 bytes per slot, only on shapes whose stage 2 reads them:
 * mfar_set_stage2_dump) and, on a bf16 index that has run a repair, the exhaustive chain pass's score block (up to 64 queries x rows x 4
 * bytes per slot; sized down when HBM is short).  Other per-launch scratch (lists, candidate tables: MBs) is not counted.  Any pointer
 * may be NULL.  No reference counterpart. */
int mfar_index_resident_bytes(const mfar_index* idx, int64_t* rows, int64_t* screen, int64_t* gather, int64_t* tables, int64_t* dumps);

/*
 * Write n row-major fp32 vectors src[n, dim] into field `field`, local rows [local_row0, local_row0 + n).
 * Replaces MemoryMapDict.__setitem__ (data/util.py:40-41) as driven by on_eval_start (contrastive.py:482-490)
 * and the external re-assignment of DenseFlatIndex.vectors (contrastive.py:494).
 * Ordering: the write runs on `stream`, behind every launch that mfar_pipeline_* has enqueued over this index so far (the library makes
 * `stream` wait for its own streams: those launches read the OLD rows to the end); every search / launch / read_rows issued after the
 * call sees the new rows, whatever stream it runs on (an asynchronous write leaves an event behind it that the next reader waits for).
 * Callers that drive the split-phase entry points (mfar_stage1_begin / _finish, mfar_search_stage2) on streams of their own order
 * their writes behind the batches THEY still have in flight themselves.
 */
int mfar_index_write_rows(mfar_index* idx, int field, int64_t local_row0, int64_t n, const float* src, int on_device,
                          void* stream);
/* Read rows back as row-major fp32 == the reference memmap layout (raw [D,E] float32, data/util.py:35). */
int mfar_index_read_rows(mfar_index* idx, int field, int64_t local_row0, int64_t n, float* dst, int on_device,
                         void* stream);

/*
 * Stage 1 == DenseFlatIndex.retrieve_batch (data/index.py:181-222) for ALL fields in one pass
 * (the loop at contrastive.py:672-674): per (query, field) exhaustive top-k over the shard's rows.
 *   q            [Q, dim] fp32 query embeddings (index.py:184-185 ndarray branch)
 *   sentinel     != 0: lists are seeded with k x (doc 0, score 0.0) as index.py:192-193 does, so only strictly
 *                positive scores enter and short lists are padded with (0, 0.0); == 0: no seed, padding (-1,-inf)
 *   field_ids    [Q, n_fields, k] int64 global ids, field_scores [Q, n_fields, k] fp32, canonical order
 */
int mfar_retrieve_fields(mfar_index* idx, const float* q, int Q, int k, int sentinel, int64_t* field_ids,
                         float* field_scores, int on_device, void* stream);

/*
 * Stage 1 for ONE field == one DenseFlatIndex.retrieve_batch call of the reference's per-field loop
 * (data/index.py:181-222 as driven by contrastive.py:672-674): scans only that field's rows.  ids/scores [Q, k].
 */
int mfar_retrieve_field(mfar_index* idx, int field, const float* q, int Q, int k, int sentinel, int64_t* ids, float* scores,
                        int on_device, void* stream);

/*
 * Stage 2 == DenseFlatIndex.score_batch (data/index.py:227-232) for all fields (contrastive.py:681-683):
 * out[Q, C, n_fields] = <q_i, slab[f, cand[i,c]]>.  cand [Q, C] int64 global ids; ids outside this shard
 * (or < 0) produce NaN (the reference raises KeyError for unknown keys -- the Python wrapper keeps that).
 */
int mfar_score_candidates(mfar_index* idx, const float* q, int Q, const int64_t* cand, int C, float* out,
                          int on_device, void* stream);

/*
 * Mixer == mask (contrastive.py:686) + LinearWeights.forward (modeling/weighting.py:17-29) + topk (contrastive.py:696).
 *   cand_scores [Q, C, F], cand_ids [Q, C] (int64, < 0 = empty slot), n_cand [Q] or NULL (= C everywhere)
 *   q [Q, E]; W [E, F] when query_cond (weight = q @ W) else [F] (weighting.py:24-27); mask [F] or NULL
 *   ids/scores [Q, k]; n_valid [Q] = min(#candidates, k) (the reference's topk raises when fewer than k
 *   candidates exist; rows past n_valid are (-1, -inf)).   C <= 4096, F <= MFAR_MAX_FIELDS.
 */
int mfar_mix_topk(int device, const float* cand_scores, const int64_t* cand_ids, const int32_t* n_cand, const float* q,
                  const float* W, int query_cond, const float* mask, int Q, int C, int F, int E, int k, int64_t* ids,
                  float* scores, int32_t* n_valid, int on_device, void* stream);

/*
 * The whole per-batch scorer == RetrievalTrainingModule.trec_eval_step (modeling/contrastive.py:669-704) with one
 * embedding per query: stage 1 -> union of ids (:678-679) -> stage 2 -> mask -> field-weight softmax -> top-k2.
 * Optional outputs (may be NULL): field_ids/field_scores [Q, F, k1] and n_cand [Q].
 * Requires n_fields * k1 <= 4096.
 */
int mfar_search_two_stage(mfar_index* idx, const float* q, int Q, const float* W, int query_cond, const float* mask,
                          int k1, int k2, int sentinel, int64_t* ids, float* scores, int32_t* n_valid,
                          int64_t* field_ids, float* field_scores, int32_t* n_cand, int on_device, void* stream);

/*
 * Fused mode (no reference counterpart): the EXHAUSTIVE mix instead of the two-stage one.  The reference scores a
 * document as sum_f softmax_f(q W) mask_f <q, d_f> (weighting.py:25-29, contrastive.py:685-694) but only over the union of the
 * per-field top-100 lists; folding the gate into the query, that sum is one inner product in n_fields * dim dimensions, and
 * this entry point returns its top-k over ALL rows of the shard: ids / scores [Q, k] (padding (-1, -inf)), canonical order.
 * It is a different result set than mfar_search_two_stage (a document strong in the mix but outside every per-field list is
 * found here and not there): it claims Recall@k parity, not id parity, and bench.py reports it separately.  The score of a
 * pair is the contract's fma chain over the n_fields * dim folded products.  The first call builds a companion slab of rows of
 * dim n_fields * dim over the same documents (+100 % HBM for the fp32 copy); fp32 indexes only, k < 128.  It is scanned by
 * the exact fp32 MFMA pass: 2 * D * F * E flops per query at the fp32 MFMA rate, i.e. SLOWER than the certified two-stage path
 * on this hardware (the certificate's norm bound does not shrink with averaged scores, so screening does not pay for it).
 */
int mfar_search_fused(mfar_index* idx, const float* q, int Q, const float* W, int query_cond, const float* mask, int k,
                      int64_t* ids, float* scores, int on_device, void* stream);

/*
 * Second half of mfar_search_two_stage on its own: union (contrastive.py:678-679) -> stage 2 (:681-683) -> mask, field
 * weights, top-k2 (:685-696), given the stage-1 lists produced by mfar_retrieve_fields.  Device pointers only, nothing
 * synchronises.  `slot` (0 .. 3) selects one of four internal workspaces: with mfar_retrieve_fields(batch i+1) on one stream
 * and mfar_search_stage2(batch i) on another, two batches overlap on the GPU.
 * field_scores [Q, n_fields, k1] (may be NULL) + sentinel: the lists' exact scores and their padding convention, as
 * mfar_retrieve_fields / mfar_stage1_finish wrote them.  When given, a candidate's score in the field whose list it came from is
 * taken from the list instead of being gathered again (stage 1 and stage 2 walk the same fma chain: identical bits).
 */
int mfar_search_stage2(mfar_index* idx, const float* q, int Q, const float* W, int query_cond, const float* mask, int k1,
                       int k2, const int64_t* field_ids, const float* field_scores, int sentinel, int slot, int64_t* ids,
                       float* scores, int32_t* n_valid, int32_t* n_cand, void* stream);
/*
 * The same for a SWEEP of field masks (mask_fields.py:143-170 evaluates baseline + one masked run per field / field type /
 * field name: 2 F + 2 runs that differ in the mask only): masks [n_masks, F]; the candidate union and stage 2 run once, the
 * mixer once per mask.  ids / scores [n_masks, Q, k2], n_valid [n_masks, Q] (may be NULL); device pointers.
 */
int mfar_search_stage2_masks(mfar_index* idx, const float* q, int Q, const float* W, int query_cond, const float* masks,
                             int n_masks, int k1, int k2, const int64_t* field_ids, const float* field_scores, int sentinel, int slot,
                             int64_t* ids, float* scores, int32_t* n_valid, int32_t* n_cand, void* stream);

/*
 * Multi-GPU (row shards + one exchange, replaces the file-based exchange of contrastive.py:491-494,519-536):
 *   mfar_search_local  : stages 1+2 on this shard -> a fixed-size payload (mfar_payload_bytes) holding the shard's
 *                        per-field lists and the F-score vector of every local candidate;
 *   mfar_merge_payloads: given the n_shards payloads (all-gathered by the caller over RCCL), merge per-field lists,
 *                        form the global candidate union, mix and take the final top-k2.  Every rank computes the
 *                        same answer.  With n_shards == 1 the result equals mfar_search_two_stage.
 */
int64_t mfar_payload_bytes(int Q, int n_fields, int k1);
/* phases: 1 = header + stage 1 (per-field lists), 2 = local union + stage 2 (reads the lists from the payload),
 * 3 = both.  Split phases (device buffers only) let the caller run phase 2, the all-gather and the merge of batch i on a
 * side stream while phase 1 of batch i+1 runs on the main stream. */
int mfar_search_local(mfar_index* idx, const float* q, int Q, int k1, int sentinel, void* payload, int phases,
                      int on_device, void* stream);
/* workspace: optional caller-owned device scratch of mfar_merge_workspace_bytes() bytes (then the call never
 * synchronises the stream); NULL = internal scratch shared per device, serialised and synchronised. */
int64_t mfar_merge_workspace_bytes(int Q, int n_fields, int k1);
int mfar_merge_payloads(int device, const void* payloads, int n_shards, const float* q, int Q, int E, const float* W,
                        int query_cond, const float* mask, int n_fields, int k1, int k2, int sentinel, int64_t* ids,
                        float* scores, int32_t* n_valid, void* workspace, int64_t workspace_bytes, int on_device,
                        void* stream);

/*
 * Lists-first exchange: the multi-GPU path for more than a few ranks.  Two SMALL collectives replace the one large
 * payload: (1) every rank all-gathers only its stage-1 lists (mfar_retrieve_lists -> mfar_lists_bytes() bytes);
 * (2) mfar_search_owned merges the gathered lists, forms the same global candidate union on every rank, re-scores and
 * mixes ONLY the candidates whose rows this rank owns and writes its local top-k2 (mfar_topk_bytes() bytes); the ranks
 * all-gather those and mfar_merge_topk takes the final top-k2.  Per rank, stage 2 then costs ~1/n_shards of the
 * single-payload scheme, and the bytes on xGMI drop from ~2.7 MB to ~0.7 MB per rank per batch.  Same results, bit for
 * bit, as mfar_search_two_stage on the unsharded corpus.  Device pointers only; nothing synchronises; `slot` as above.
 */
int64_t mfar_lists_bytes(int Q, int n_fields, int k1);
int64_t mfar_topk_bytes(int Q, int k2);
int mfar_retrieve_lists(mfar_index* idx, const float* q, int Q, int k1, int sentinel, void* lists, void* stream);
/* any_fail (device int32, may be NULL = 0): this rank's certificate flag of the batch as reported by mfar_stage1_finish; it
 * travels inside the top-k payload, and mfar_merge_topk writes the OR over all ranks to *any_fail (device int32, may be
 * NULL) -- every rank takes the same redo decision without a third collective. */
int mfar_search_owned(mfar_index* idx, const void* gathered_lists, int n_shards, const float* q, int Q, const float* W,
                      int query_cond, const float* mask, int k1, int k2, int sentinel, int slot, const int32_t* any_fail,
                      void* topk, void* stream);
/* The same for a sweep of field masks [n_masks, F] (see mfar_search_stage2_masks): everything up to the owned candidates' score
 * vectors once, then one local top-k payload per mask: topk holds n_masks payloads of mfar_topk_bytes() each, back to back. */
int mfar_search_owned_masks(mfar_index* idx, const void* gathered_lists, int n_shards, const float* q, int Q, const float* W,
                            int query_cond, const float* masks, int n_masks, int k1, int k2, int sentinel, int slot,
                            const int32_t* any_fail, void* topk, void* stream);
int mfar_merge_topk(int device, const void* gathered_topk, int n_shards, int Q, int k2, int64_t* ids, float* scores,
                    int32_t* n_valid, int32_t* any_fail, void* stream);

/* Stream choreography helper for pipelined batches: make `stream` wait until the most recently enqueued FULL stage-1
 * kernel of this handle is about to start (i.e. until everything enqueued before it, including the sample pass, has
 * finished).  Work enqueued on `stream` afterwards (the light tail kernels of the previous batch) then runs BESIDE that
 * long MFMA-bound kernel instead of delaying its start. */
int mfar_stream_wait_stage1_start(mfar_index* idx, void* stream);

/*
 * Split-phase stage 1 for pipelined callers (device pointers, Q <= mfar_max_split_batch(), asynchronous on `stream`).
 *   mfar_stage1_begin   query preparation, sample pass, the long scan and the list merge of batch `slot` (four slots,
 *                       0 .. 3: a caller keeps as many launches in flight as it uses slots; mfar.data.pipeline uses three).
 *                       Without the fp16 screen this already leaves the final lists in field_ids / field_scores.
 *   mfar_stage1_finish  with the screen: exact re-scoring of the min(k + 92, 192) screened rows per list and the certificate ->
 *                       field_ids / field_scores [Q, n_fields, k], exactly what mfar_retrieve_fields returns; without:
 *                       nothing.  May run on another stream than begin (the caller orders finish after begin with an
 *                       event) and beside the begin of the OTHER slot -- its kernels are small enough to be resident next
 *                       to the scan.  A slot may be begun again once its finish has completed.
 *   any_fail            NULL: a failed certificate is repaired inside finish (the exact fp32 pass is launched and idles
 *                       when nothing failed).  Non-NULL (device int32): finish only reports -- *any_fail != 0 means the
 *                       lists of this batch are NOT final and the caller must redo the batch with mfar_retrieve_fields
 *                       (which repairs failed fields itself).  This keeps the fp32 kernel, which cannot be co-resident
 *                       with the other slot's scan, off the stream in the common case.
 * Same arguments (q, Q, k, sentinel, outputs) must be passed to both calls of a batch.
 */
int mfar_stage1_begin(mfar_index* idx, const float* q, int Q, int k, int sentinel, int slot, int64_t* field_ids,
                      float* field_scores, void* stream);
/*
 * The WIDE screened pass.  The screened scan is HBM-bound with the MFMA pipe at ~40 %: two fp16 query terms x 64 queries.
 * A block of 65 .. 128 queries is scanned with ONE fp16 term per query instead (mfar_stage1_f16w_kernel): the same MFMA
 * work per byte, the screen slab read once per 128 queries instead of once per 64.  The certificate's error bound includes
 * the query rounding (about 2x the two-term bound); outputs stay bit-identical to the exhaustive fp32 pass, lists whose
 * proof fails are redone exactly as before.  Every entry point that takes Q queries cuts them into blocks of 128 (wide)
 * while more than 64 remain, so callers with larger batches (the reference's dev_batch_size is 64, train.py:45; pipelined
 * callers may coalesce two batches: mfar.data.pipeline) halve the scan bytes per query.
 *   mfar_max_split_batch  queries one mfar_stage1_begin / finish batch may hold: 128 when the wide pass is available for
 *                         this index and depth k (builds the screen slab if it is not current), else 64.
 *   mfar_set_wide         0: never use the wide pass (every block is 64 queries); default 1.  Environment: MFAR_WIDE.
 */
int mfar_max_split_batch(mfar_index* idx, int k);
int mfar_set_wide(mfar_index* idx, int enable);
/*
 * How the exact pass REPAIRS fields whose certificate failed (the entry points that repair on the device launch it behind
 * every screened block; it is idle when nothing failed).  A repair walks a finely cut table -- every field up to a whole
 * wave of chunks, scanned by ONE wave of workgroups -- so that a single failed field is scanned by the whole GPU (1 M x 8:
 * under 1 ms instead of the several ms its share of the grid would take) while an idle repair stays three short launches.
 *   thorough = 0 (default)  no sample pass: right when failures are rare;
 *   thorough = 1            the repair runs its own sample pass (thresholds for the failed fields: 20 % faster when most
 *                           fields fail, two more launches when none does).  For callers that only ask for repairs after a
 *                           failure was reported (mfar.data.pipeline.PipelinedSearcher sets it).
 */
int mfar_set_repair_mode(mfar_index* idx, int thorough);
int mfar_stage1_finish(mfar_index* idx, const float* q, int Q, int k, int sentinel, int slot, int64_t* field_ids,
                       float* field_scores, int32_t* any_fail, void* stream);

/* Instrumentation used by bench.py: when enabled, every stage-1 kernel launch on this handle is bracketed by HIP
 * events recorded on the stream it is launched on.  mfar_set_timing(idx, 1) enables and resets the counters;
 * mfar_stage1_timing() synchronises the recorded events and returns their summed duration and the launch count. */
int mfar_set_timing(mfar_index* idx, int enable);
int mfar_stage1_timing(mfar_index* idx, double* total_ms_out, int* n_launches_out);
/* Name of the scan kernel the most recent stage-1 launch of this handle ran (the one the timing events bracket): which of the
 * instantiations was chosen depends on the dtype, the dim, the block width, ROW MODE and AUTO-OFF.  "" before the first launch. */
const char* mfar_last_stage1_kernel(const mfar_index* idx);
/* Tunable: workgroups per CU for the stage-1 kernel (default 2). */
int mfar_set_wgs_per_cu(mfar_index* idx, int wgs);

/*
 * Certified fp16 screening of an fp32 index (no reference counterpart; the outputs of every entry point above are
 * bit-identical with and without it).  Stage 1 of an fp32 index is bound by the fp32 MFMA rate; with the screen it
 * scans an fp16 copy of each field's UNIQUE rows (bit-identical rows -- documents that share a text, above all "" for a
 * missing field, format.py:58-59 -- are scanned once) for the min(k + 92, 192) best approximate scores per (query, field),
 * re-scores those rows exactly from the fp32 slab, expands them to their documents, and PROVES from a rigorous error bound
 * that no other row can enter or tie into the exact top-k; a field whose proof fails is re-done by the exact fp32 pass over
 * the documents on the device (csrc/mfar_screen.h).  The screen slab (at most +50 % HBM) is built lazily by the first search
 * after rows were written.
 *   mode      0 = off, 1 = auto (default; indexes with >= 16384 rows, dim <= 2560, k <= 128), 2 = whenever the shapes allow.
 *             Environment default: MFAR_SCREEN.
 *             A bf16 index takes the same certified stage 1 in the same modes WITHOUT a second copy of its rows: the scan reads the
 *             bf16 slab itself (blocks of <= 64 queries: two bf16 query terms; 65 .. 128: the docs converted to fp16 in registers
 *             against one fp16 term), ranks unique rows through one bit per row, re-scores from the row-major bf16 companion.  Its
 *             certified lists equal the exact natural-order fp32 chain over the bf16 rows bit for bit; see "bf16 contract" below for
 *             the lists that are not certified.
 *   bf16 contract: which bits a bf16 index returns, and when.  The contract of a bf16 index is the natural-order fp32 fma chain over the
 *             stored bf16 values (what stage 2 and the mixer always compute).  With the certified stage 1 in use -- modes 1 and 2, ANY
 *             number of rows (the scan reads the slab itself, there is nothing to allocate), dims whose k-steps (dim / 16) divide by 4 or
 *             6 -- EVERY per-field list carries exactly those bits: a list is either certified, or written by the exhaustive chain pass
 *             (csrc/mfar_exact16.h: the same chain on the VALUs for every document of the field, ~30x the time of the screened scan, run
 *             only for fields whose certificate failed or that AUTO-OFF has switched off).  Results therefore do not depend on how the
 *             rows are sharded, on which lists happened to fail, or on the batch composition.  The plain three-term MFMA pass sums the
 *             same exact products in the matrix core's own order and agrees with the chain to ~1e-4: it is what answers in mode 0, for
 *             the other dims, and when the unique-row tables cannot be allocated -- there list members at a near-tied cut-off and the
 *             lists' score bits may differ from the chain's (final scores of common documents stay bit-identical: stage 2 recomputes them).
 *   eps_mult  multiplies the error bound of the proof; 1 = rigorous.  Test knob: a huge value makes every proof fail
 *             (exercises the exact fall-back), 0 disables the proof (NOT exact any more).
 * mfar_screen_stats synchronises the device: built = the screen slab is current, screen_bytes = its size,
 * n_checked / n_failed = (query, field) lists certified / sent to the exact fall-back since the handle was created.
 */
int mfar_set_screen(mfar_index* idx, int mode, float eps_mult);
int mfar_get_screen(const mfar_index* idx, int* mode, float* eps_mult);
int mfar_screen_stats(mfar_index* idx, int* built, int64_t* screen_bytes, int64_t* n_checked, int64_t* n_failed);
/* After the screen was built: the number of distinct vectors of a field (what the screened pass scans) and the size of its
 * largest group of bit-identical rows; -1 while no screen is current. */
int mfar_screen_field_info(mfar_index* idx, int field, int64_t* n_unique_rows, int64_t* largest_group);

/*
 * Certified two-level stage 2 (no reference counterpart; the outputs of every entry point above are bit-identical with and
 * without it).  Stage 2 == DenseFlatIndex.score_batch x F (data/index.py:227-232, contrastive.py:681-683) gathers one row per
 * (candidate, field) pair, and the candidate union grows with the number of fields: F^2 * k1 rows per query.  With the gather
 * slab -- a row-major 16-bit copy of every row, built with the screen: fp16 of the centred + scaled values for an fp32 index
 * (+50 % HBM), the bf16 values themselves for a bf16 index (+100 %, whole-line gathers instead of 32-byte segments) -- an fp32
 * index scores every pair APPROXIMATELY at half the bytes, bounds every candidate's mixed score (contrastive.py:685-694) from both
 * sides with a rigorous error bound evaluated through the mixer's own fma chain, and gathers fp32 rows only for the candidates
 * whose upper bound reaches the k2-th largest lower bound (csrc/mfar_select.h: mfar_s2_prune_kernel); those are mixed exactly as
 * before, so ids and score bits of the top-k2 do not change.
 *   mode   0 = gather every (candidate, field) row from the fp32 slab, 1 = two-level when available (default; fp32 index whose
 *          screen and gather slab are current, more candidates than k2, at most two masks in the call: the bounds cost one
 *          selection per mask, which a sweep of 2 F + 2 masks does not earn back), 2 = also for sweeps of any number of masks
 *          (the survivors are the union over the masks).  Environment default: MFAR_STAGE2_PRUNE;
 *          MFAR_GATHER_SLAB=0 never allocates a gather slab.  The error bound is scaled by mfar_set_screen's eps_mult (test knob).
 * mfar_stage2_stats synchronises the device: candidates the prune kernel has seen / survivors it kept since the handle was created.
 */
int mfar_set_stage2_mode(mfar_index* idx, int mode);
/* Which kernels run the tail of a search (union -> stage 2 -> mixer); results are identical.  1 (default; environment MFAR_S2_FUSED): the
 * round-6 family -- field weights / q . mean / eps once per batch and off the critical chain (mfar_s2_gate_kernel), candidate union + known
 * pairs by bitmap in one kernel when the ids span <= 2^20 (mfar_s2_front_kernel), the interval bounds by several workgroups per query
 * (mfar_s2_bounds_kernel + mfar_s2_select_kernel);  0: the one-workgroup-per-query kernels of rounds 3-5 (sort-based union, prep, known,
 * prune).  A diagnostic and a test hook (tests/test_gpu_stage2.py runs both against each other and the oracle). */
int mfar_set_stage2_kernels(mfar_index* idx, int family);
/*
 * SCORE DUMP (no reference counterpart; outputs bit-identical with and without it).  The approximate level of the two-level stage 2
 * normally gathers one 16-bit row per (candidate, field) pair.  When queries x candidates exceeds a field's rows -- many fields, small
 * corpora, the row shards of a multi-GPU run -- the wide screened pass of stage 1 instead WRITES every approximate score it computes
 * (16-bit codes: rows x 128 queries x 2 bytes per field and launch, per pipeline slot) and stage 2 reads its pairs out of that table,
 * bounded by the screened pass's own eps + the quantisation step (csrc/mfar_select.h: mfar_s2_lookup_kernel): 129 375 x 22 moves
 * 0.7 + 0.5 GB instead of 7.9 GB per 128 queries.
 *   mode   0 = never, 1 = when the dump moves less than ONE THIRD of the gathers' bytes (default; environment MFAR_S2_DUMP),
 *          2 = whenever the wide pass of an fp32 index with a current screen runs.
 * mfar_stage2_dump_info: would a launch with list depth k1 use it; bytes one launch writes; launches that read it so far.
 */
int mfar_set_stage2_dump(mfar_index* idx, int mode);
/*
 * ROW MODE of the certified screen (no reference counterpart; outputs bit-identical with and without it).  The certificate bounds every
 * unscanned row of a field with the field's LARGEST centred row norm; one outlier row -- or a heavy-tailed field -- then makes lists fail
 * that a per-row bound would prove.  A field whose largest norm exceeds 1.5 x the mean norm is ELIGIBLE: its wide screened pass can rank
 * rows by approx + eps(row norm), an upper bound of the exact score, and the certificate then needs no field-wide norm (csrc/mfar_screen.h
 * "ROW MODE").  It costs the scan ~7 % (its own kernel instantiation), so it is switched on only when it pays:
 *   mode   0 = never, 1 = auto (default; environment MFAR_SCREEN_ROW_MODE): eligible fields are activated by mfar_row_mode_activate, which
 *          mfar.data.pipeline.PipelinedSearcher calls when a launch reports a failed certificate; 2 = eligible fields are always active.
 * mfar_row_mode_info: bit f of the masks = field f is eligible / active (valid once the screen is built).
 */
int mfar_set_row_mode(mfar_index* idx, int mode);
/*
 * AUTO-OFF and inline repair: the worst case of the certified screen (no reference counterpart; outputs bit-identical in every state).
 * A certificate fails when more than k' - k unique rows of a field score within ~2 eps of the list's k-th best -- clusters of
 * near-duplicate rows (not bit-identical, so the unique-row build keeps them apart) do that list after list, and every failure costs the
 * exact pass of that field on top of the screen.  The library therefore reads the certificate flags of finished launches (copied to
 * pinned host memory behind the certify kernel, picked up by a later call; nothing waits) and decides per FIELD: a field that failed in
 * >= off_fails of its last 16 screened launches is switched OFF -- the screen leaves it out and the exact fp32 pass writes its lists
 * directly, on the scan stream; every probe_every-th launch screens it anyway (quietly) and two clean probes in a row switch it back on.
 * With every field off a launch is the exact pass and nothing else, i.e. the screen can cost a hostile corpus the probes (< 1 %), not 2x.
 * Independently, when >= 4 of the last 16 launches reported a failure among the fields that are on, mfar_stage1_finish repairs on the
 * device even when asked to report only (and reports a clean batch); it returns to reporting when at most one of the last 16 failed.
 *   mode         0 = never switch a field off, 1 = auto (default; environment MFAR_SCREEN_AUTO_OFF).  fp32 indexes as described; a bf16
 *                index switches a field off only after 16 failed launches of 16 (its exact pass is the VALU chain pass, ~30x a screened
 *                scan of the field: what switching off saves there is the field's share of the screened scan and its certificate).
 *   off_fails    1 .. 16 failed launches of the last 16 (0 = keep; default 12: a field that fails less often is cheaper screened + repaired)
 *   probe_every  launches between probes (0 = keep; default 64)
 * mfar_auto_off_info: bit f of off_fields = field f is off now; fields switched off / back on, probe launches so far; whether finish
 * currently repairs inline.  Any pointer may be NULL.
 */
int mfar_set_auto_off(mfar_index* idx, int mode, int off_fails, int probe_every);
/*
 * TIER 2 of the certified screen: the THRESHOLD RESCAN (no reference counterpart; outputs bit-identical in every mode).  A list whose first
 * certificate fails has still produced e_k, the exact k-th best score among its re-scored rows -- a lower bound of the true k-th best, so every
 * row of the exact top-k has approximate score >= e_k - eps.  Instead of sending the field to the exact fp32 pass (bound by the fp32 MFMA
 * rate: ~7x its screened scan), the library rescans the fields that hold failed lists with the same screened kernel over the same fp16
 * rows and that FIXED threshold per list, gathers every row above it from the fp32 slab (a few hundred per list on encoder-produced and
 * near-duplicate corpora: profiles/r06_tier2_population.txt), and takes the exact top-k of that complete set -- no second proof needed.
 * Lists that need more than 2048 candidates (or overflow a chunk list) go to the exact pass as before; AUTO-OFF and the inline-repair
 * decision see the flags AFTER tier 2.  All-fields searches; fp32 indexes, and bf16 indexes (the rescan reads the bf16 slab itself, the
 * candidates are re-scored from the row-major companion with the natural-order chain: the lists tier 2 finishes carry the bits of the
 * "bf16 contract" above, and the ~30x chain pass is left to the lists that overflow).
 *   mode   0 = never, 1 = auto (default; environment MFAR_SCREEN_TIER2): its kernels follow a certificate only while a launch of the
 *          last 256 had a failed first certificate -- a corpus whose lists all certify never pays their (idle) launches; 2 = always.
 * mfar_tier2_stats synchronises the device: whether tier 2 is armed now; lists handed to it / lists it had to pass on to the exact pass
 * since the handle was created; causes [4] = why: a chunk list of the (re)scan reached its depth, more than 8192 rows above the threshold,
 * more than 2048 candidates in the band, ties across the cut.  Any pointer may be NULL.
 * The rescan is the FALLBACK (ABI 107): the launch's own scan appended every row scoring at least its sample threshold tg(q, f) to its
 * chunk lists, and tg -- the k'-th best of a few percent of the rows -- usually lies below the list's tier-2 threshold T; where tg <= T and
 * no chunk list of the list was ever compacted (both checked on the device, per list), the candidates are the entries >= T of the lists
 * that scan already wrote and nothing is scanned twice.  mfar_tier2_rescan_stats (synchronises): lists whose candidates came from the
 * launch's own scan / lists that needed the rescan, since the handle was created.  In mode 1 the rescan itself is enqueued only while a
 * list of the last 256 launches asked for it (a list that asks while it is not armed goes to the exact pass and arms it); mode 2 always
 * enqueues it.  mode + 4 (or environment MFAR_T2_FIRST_SCAN=0): every list takes the rescan (diagnostic: the fallback path on demand).
 */
int mfar_set_tier2(mfar_index* idx, int mode);
int mfar_tier2_rescan_stats(mfar_index* idx, int64_t* n_from_scan, int64_t* n_rescanned);
/*
 * DEEP SCAN: tier 2 without the first attempt, for fields whose first certificates keep failing (no reference counterpart; outputs
 * bit-identical in every mode).  Such a field paid its screened scan twice (first attempt + rescan) when this was built -- tier 2 has
 * since learned to take its candidates from the first attempt's own chunk lists, which removes the reason.  Once the library has seen a field fail
 * its first certificate in 8 of its last 16 launches it stops certifying it: the ONE scan of the field runs with a complete-set threshold
 * taken from the sample pass -- T = (k-th largest sampled score) - 2 eps: k rows score at least the k-th sampled score, so the true k-th best
 * exact score is at least that minus eps and every row of the exact top-k scores approximately >= T -- its chunk lists then hold every
 * such row (a few thousand), mfar_t2_collect_kernel narrows them to the band around the k-th best approximate score of that complete set
 * (a few hundred), and tier 2's back half (exact re-scoring, selection, expansion to documents) writes the lists.  Lists that overflow go
 * to the exact pass -- and their field is demoted at once (auto mode); after 1024 launches a deep field is evaluated afresh.  Needs tier 2
 * (mfar_set_tier2 != 0), an fp32 index, and a shape whose scan runs the light sample pass.
 * OFF by default: measured it is 11 % slower than certificates on rows that certify, and on clustered rows the
 * sample-derived threshold is too loose -- 5 % of the lists hold more than 8192 rows above it, and ONE overflowing list costs its field
 * the exact pass -- so tier 2 (26 k q/s on the hostile corpus with the rescan, 40 k without) beats it (16 k).  Kept as a measured, tested
 * alternative.
 *   mode   0 = never (default; environment MFAR_SCREEN_DEEP), 1 = auto, 2 = every field, always (a test / experiment setting).
 * mfar_deep_scan_info: bit f of deep_fields = field f runs as a deep field now; fields switched to it so far.
 */
int mfar_set_deep_scan(mfar_index* idx, int mode);
int mfar_deep_scan_info(mfar_index* idx, uint32_t* deep_fields, int64_t* n_switched);
int mfar_tier2_stats(mfar_index* idx, int* armed, int64_t* n_lists, int64_t* n_passed_on, int64_t* causes);
int mfar_auto_off_info(mfar_index* idx, uint32_t* off_fields, int64_t* n_switched_off, int64_t* n_switched_on, int64_t* n_probes,
                       int* inline_repair);
int mfar_row_mode_activate(mfar_index* idx);
int mfar_row_mode_info(const mfar_index* idx, uint32_t* eligible_fields, uint32_t* active_fields);
int mfar_stage2_dump_info(mfar_index* idx, int k1, int* wanted, int64_t* bytes_per_launch, int64_t* n_launches);
int mfar_stage2_stats(mfar_index* idx, int* two_level_available, int64_t* gather_slab_bytes, int64_t* n_candidates,
                      int64_t* n_survivors);

/*
 * The batch PIPELINE behind the C ABI: what a host in any language calls to reach the rate bench.py reports (the synchronous
 * mfar_search_two_stage above scans the slab once per call and leaves the GPU idle between its kernels).  Stands behind the reference's
 * evaluation loop -- test_step per batch of dev_batch_size = 64 queries, modeling/contrastive.py:553-563 -> trec_eval_step :669-704 -- for ONE
 * row shard (the row-sharded exchange needs two collectives per launch, which stay with the host: mfar/data/pipeline.py).
 *   - `depth` launches in flight (2 .. 4, 0 = 3) on the library's own HIP streams: the scans of consecutive launches back to back on a
 *     high-priority stream, everything after a scan (exact re-scoring + certificate, union, stage 2, mixer) on side streams beside the next scan;
 *   - COALESCING: when mfar_max_split_batch() is 128, `coalesce` (0 = auto: 2) consecutive batches of max_batch <= 64 queries are scanned
 *     by one launch of the wide pass -- half the scan bytes per query; results per query are unchanged;
 *   - a launch whose certificate failed is redone exactly when its result is taken; data on which that keeps happening is handled by the
 *     library (mfar_set_auto_off);
 *   - a launch that could not be enqueued (submit / flush / result returned MFAR_ERR_NOMEM or MFAR_ERR_HIP from inside the launch) keeps its
 *     batches: their tickets stay registered (*ticket is written before the launch is attempted) and the launch is run, synchronously, when
 *     one of its results is taken or its slot comes round again -- or the error repeats there.  Nothing half-done is ever returned.
 * Results are bit for bit those of mfar_search_two_stage(idx, q, ...) for the same queries.
 *   create     W [E, F] (query_cond) or [F]; mask [F] or NULL (ones); copied (host or device pointers per on_device).
 *   submit     q [Q, E], 1 <= Q <= max_batch; host pointer (copied before the call returns) or device pointer (copied on `stream`, which
 *              also orders the copy after the caller's producer).  Returns at once; *ticket identifies the batch.
 *   result     waits for the batch's launch (launching it alone if it is still held for coalescing), copies ids / scores [Q, k2] and
 *              n_valid [Q] (may be NULL) to host memory (synchronous) or device memory (on `stream`).  A ticket stays valid until
 *              `depth` LATER launches have started.  A launch starts when it holds `coalesce` batches -- so, left alone, a ticket is
 *              valid until depth * coalesce more batches were submitted: to keep the pipeline full take results lag = depth * coalesce
 *              - 1 submissions late (mfar_pipeline_info) -- or EARLY, with what it holds, on flush / set_weights / the result of a batch
 *              it still holds: a caller who cuts launches short counts launches, not submissions.  Results of tickets submitted
 *              before a set_weights are those of the weights they were submitted under.
 *   result_view  device pointers into the launch's slot instead of copies (+ the batch's stage-1 lists [Q, F, k1]); same validity.
 *   set_weights  flushes, waits for the launches in flight and replaces W / mask (a mask_fields sweep, a new weight version).
 *   flush      launches a batch that is being held for coalescing.
 * Calls on one pipeline (and on its index) must be serialised by the caller.  The index must outlive the pipeline.
 */
typedef struct mfar_pipeline mfar_pipeline;
int mfar_pipeline_create(mfar_pipeline** out, mfar_index* idx, const float* W, int query_cond, const float* mask, int k1, int k2, int sentinel,
                         int max_batch, int depth, int coalesce, int on_device);
void mfar_pipeline_destroy(mfar_pipeline* p);
int mfar_pipeline_info(const mfar_pipeline* p, int* depth, int* coalesce, int* queries_per_launch, int* lag, int64_t* n_redone);
int mfar_pipeline_set_weights(mfar_pipeline* p, const float* W, const float* mask, int on_device);
int mfar_pipeline_submit(mfar_pipeline* p, const float* q, int Q, int on_device, void* stream, int64_t* ticket);
int mfar_pipeline_flush(mfar_pipeline* p);
int mfar_pipeline_result(mfar_pipeline* p, int64_t ticket, int64_t* ids, float* scores, int32_t* n_valid, int on_device, void* stream);
int mfar_pipeline_result_view(mfar_pipeline* p, int64_t ticket, const int64_t** ids, const float** scores, const int32_t** n_valid,
                              const int64_t** field_ids, const float** field_scores);
/* the batch's stage-1 lists [Q, n_fields, k1] copied out like mfar_pipeline_result does for the final lists */
int mfar_pipeline_lists(mfar_pipeline* p, int64_t ticket, int64_t* field_ids, float* field_scores, int on_device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MFAR_HIP_H */
