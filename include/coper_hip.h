/*
 * coper_hip.h -- C ABI of libcoper_hip.so: the MI355X (gfx950) CoPER-ConvE scoring engine.
 *
 * The reference (otiliastr/coper, CoPER_ConvE/qa_cpg) has NO FFI / plugin interface for this
 * path: the seam is a Python object protocol between run_cpg.py / metrics.py and models.py.
 * Each entry point below therefore cites the reference *Python* interface it replaces
 * (paths relative to CoPER_ConvE/qa_cpg/).  The reference-side binding a maintainer would add
 * is the ctypes stub shown in INTEGRATION.md (it is what coper_amd/_lib.py does).
 *
 * Conventions
 *   - plain C, no C++ exceptions cross the boundary; every call returns a coper_status.
 *   - all tensor arguments are DEVICE pointers owned by the caller (row-major, fp32 unless
 *     stated); ids are int64 like the reference batch contract (models.py:139-152), lookup
 *     indices int32.
 *   - every call is asynchronous on the hipStream_t it is given (passed as void* so that the
 *     header needs no HIP include); the handle's workspace is stream-ordered: use one handle
 *     per (GPU, stream).  A handle is not thread-safe.
 *   - parameters are BORROWED: the caller keeps the device buffers alive and must call
 *     coper_prepare() again after changing their contents (it rebuilds every derived cache).
 */
#ifndef COPER_HIP_H_
#define COPER_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(COPER_BUILD)
#define COPER_API __attribute__((visibility("default")))
#else
#define COPER_API
#endif

#define COPER_ABI_VERSION 3
#define COPER_MAX_CTX 8 /* max hidden layers of a g_MLP generator (models.py:44-54 loops over any number; no shipped YAML has more than 1) */

typedef enum coper_status {
  COPER_OK = 0,
  COPER_EINVAL = 1,    /* bad argument / ill-formed model_descriptors */
  COPER_EMISSING = 2,  /* a required parameter was never set */
  COPER_ESHAPE = 3,    /* parameter shape does not match the configuration */
  COPER_EHIP = 4,      /* HIP runtime error, text in coper_last_error() */
  COPER_ESTATE = 5,    /* call order violated (e.g. encode before prepare) */
  COPER_ENOMEM = 6,
  COPER_EUNSUPPORTED = 7
} coper_status;

typedef enum coper_score_mode {
  COPER_SCORE_F32 = 0,   /* entity table fp32, exact-f32 MFMA (v_mfma_f32_32x32x2_f32): parity mode */
  COPER_SCORE_BF16X3 = 1 /* "x3" (the name is round 2's): every fp32 operand split into two fp16 terms hi + lo, 3 fp16 MFMAs per
                          * product with fp32 accumulation.  Operands are first moved into fp16's window by exact powers of two
                          * (per table, per relation's weights, per packed batch of queries), so hi + lo carries 22 bits at any
                          * magnitude: logits within 2.5e-7 |h||E| of the fp32 chain; integer ranks ARE the fp32 chain's (the
                          * exact band, rank_band_kappa below), checked at run time by coper_band_audit */
  /* (a single-16-bit mode, ~2^-9 per product, cannot meet the 1e-3 logit gate of the path and is not part of the ABI) */
} coper_score_mode;

/* Mirrors the `model_descriptors` dict of ConvE.__init__ (models.py:98-130) -- only the keys
 * that influence inference -- plus what the build adds: (emb_h, emb_w) replacing the
 * hard-coded (10, d // 10) of models.py:261-262,355; the entity shard; the score precision. */
typedef struct coper_config {
  int32_t abi_version;          /* = COPER_ABI_VERSION */
  int32_t device;               /* HIP device ordinal */
  int64_t num_ent;              /* |E| (global)                          models.py:102 */
  int64_t num_rel;              /* R2 = 2|R| incl. _reverse relations    models.py:103 */
  int32_t ent_emb_size;         /* d                                     models.py:104 */
  int32_t rel_emb_size;         /* r                                     models.py:105 */
  int32_t emb_h, emb_w;         /* image reshape, emb_h * emb_w == d     models.py:355 */
  int32_t conv_filter_height;   /* default 3                             models.py:109 */
  int32_t conv_filter_width;    /* default 3                             models.py:110 */
  int32_t conv_num_channels;    /* default 32                            models.py:111 */
  int32_t concat_rel;           /*                                       models.py:113 */
  int32_t do_parameter_lookup;  /* g_lookup                              models.py:107 */
  /* context_rel_conv / context_rel_out (models.py:114-115):
   *   n = -1 -> None (static parameter), n = 0 -> [] (g_linear), n > 0 -> hidden sizes (g_MLP) */
  int32_t n_ctx_conv;
  int32_t ctx_conv[COPER_MAX_CTX];
  int32_t n_ctx_out;
  int32_t ctx_out[COPER_MAX_CTX];
  int32_t context_rel_use_batch_norm; /*                                 models.py:117 */
  float bn_epsilon;             /* 1e-3 = tf.layers.batch_normalization default */
  int64_t shard_lo, shard_hi;   /* entity rows [lo, hi) held by this handle; [0, num_ent) = unsharded */
  int32_t score_mode;           /* coper_score_mode */
  /* COPER_SCORE_BF16X3: kappa, the relative half-width of the exact band.  A comparison of a competitor's logit with the
   * target's that is closer than
   *       tau_q = 2 (kappa (|h_q| max|E_e| + 8 max|pred_bias|) + 2^-25 sqrt(d) (max|E_e| 2^-e_h + |h_q| 2^-e_E))
   * is decided by the fp32 chain of COPER_SCORE_F32 instead of the mode's own arithmetic (integer ranks: metrics.py:44-50).
   * 0 = the library default 1e-6: four to five times the largest error measured over 6e6 - 3e8 logits at every operand scale
   * (tests/test_gpu_scale.py: <= 0.24 kappa (...) ), for logits dominated by their products and by pred_bias alike (the
   * weight 8: an accumulation that runs at the magnitude of the bias rounds to ulps of it at every step).  The last term is
   * rigorous: elements more than 2^17 below their class maximum lose bits to fp16's subnormals (e_h, e_E: the powers of
   * two of the packed batch and of the table).  kappa itself is EMPIRICAL -- the proven bound of the split and of fp32
   * accumulation in any order is 3 * 2^-22 + 2 (48 ceil(d/16) + 1) 2^-24 = 7.5e-5 at d = 200 -- which is why every count
   * launch can audit it (coper_band_audit).  Larger = more pairs re-scored by the chain, proportionally. */
  float rank_band_kappa;
  /* COPER_SCORE_BF16X3: the entity planes hold ent_emb 2^e with e chosen so that this magnitude lands in [2^14, 2^15) of
   * fp16's range.  0 = the largest |ent_emb| element of the handle's own rows.  Entity shards of one table pass the table-wide
   * maximum (coper_amd/sharding.py all-reduces it) so that the mode's logits do not depend on the shard layout; a value below
   * the shard's own maximum is refused by coper_prepare.  The same magnitude bounds the encoder's INPUT rows: rows handed in through
   * `e1_rows` (an entity-sharded evaluation passes rows of other shards) must not exceed it in any element -- the conv activations'
   * power of two and the fp16 image planes of the fused encoder are derived from it; larger elements are clamped to fp16's range
   * and `h` saturates silently.  coper_amd/sharding.py keeps the table-wide maximum agreed on every chunk. */
  float x3_ent_absmax;
  /* COPER_SCORE_BF16X3: which count launches carry the band audit (coper_band_audit).  0 = the library default: the first
   * launch after coper_prepare and every 8th from there (+3 us per 0.5 ms pass), every launch of more than 2^31 logits;
   * n > 0: every n-th launch; negative: never.  Launches recorded into a hipGraph carry the audit only with n = 1. */
  int32_t band_audit_period;
  /* What the handle is built for (round 6; 0 = COPER_ROLE_BOTH = every earlier caller).  An entity-sharded evaluation runs the
   * rank's share of the encoder and its count over the rank's entity rows on TWO streams (coper_amd/sharding.py: steps 1 - 2 of
   * chunk n + 1 under chunk n's count launch), which one handle's workspace cannot serve: it creates one handle per role over the
   * SAME parameter tensors.  COPER_ROLE_ENCODE builds no entity planes (4 x |E_local| d 2 + |E_local| d 4 bytes) and refuses the
   * scoring entry points; COPER_ROLE_SCORE evaluates no generator and caches no W_r (R2 F d 4 bytes: 12.8 GB for the 10M-entity
   * config) and refuses coper_encode / coper_encode_rank / training.  Everything both roles derive (folded BN, the band constants,
   * the powers of two of split16.h) is the same values on both. */
  int32_t role;                 /* coper_role */
  /* The generated dense weights W_r of the relations r with r mod rel_mod_world == rel_mod_rank ONLY (round 6; 0 or 1 = all of
   * them).  An entity-sharded evaluation splits its encoder by relation -- rank g of G encodes the queries whose relation id is
   * g mod G (coper_amd/sharding.py step 2) -- so its encoder handle needs R2 / G weight sets, not R2: 1.6 GB instead of 12.8 GB per
   * rank for the 10M-entity config at G = 8.  Everything small that is derived per relation (conv filters, biases, the powers
   * of two of split16.h) is still built for every relation, from the same values: h[b] is the same bits as on a handle that holds
   * all weight sets.  A query whose relation the handle does not hold is encoded with another relation's weights and COUNTED
   * (coper_check_ids).  COPER_SCORE_BF16X3 with generated dense weights (context_rel_out) on the fused encoder only:
   * COPER_EUNSUPPORTED from coper_prepare otherwise; no training. */
  int32_t rel_mod_world;
  int32_t rel_mod_rank;
  int32_t reserved[1];
} coper_config;

typedef enum coper_role { COPER_ROLE_BOTH = 0, COPER_ROLE_ENCODE = 1, COPER_ROLE_SCORE = 2 } coper_role;

typedef struct coper_handle coper_handle;

/* Replaces ConvE.__init__ / _create_variables shape logic (models.py:98-130, 203-336):
 * validates the configuration exactly as the reference derives its shapes.  No parameter
 * memory is allocated: parameters are borrowed through coper_set_param. */
COPER_API int coper_create(const coper_config* cfg, coper_handle** out);
COPER_API void coper_destroy(coper_handle* h);

/* Text of the last failure on this handle (never NULL).  With h == NULL: last coper_create failure. */
COPER_API const char* coper_last_error(const coper_handle* h);
COPER_API int coper_abi_version(void);

/* CRC-32C (Castagnoli) of a host buffer, continuing from `crc` (0 for a fresh one): the checksum of the TensorFlow
 * checkpoint files `tf.train.Saver` writes and reads (run_cpg.py:189,206,252; tensor_bundle.proto crc32c fields, the table
 * blocks of the .index file).  Host code (SSE4.2 crc32 instruction when the CPU has it): coper_amd/tf_bundle.py checks a
 * 500 MB checkpoint with it in well under a second, its NumPy form needs 65 MB/s. */
COPER_API uint32_t coper_crc32c(uint32_t crc, const void* data, uint64_t n);

/* Derived sizes (models.py:261-271): F = fc_input_size, Ho/Wo = conv output; n_local = shard_hi - shard_lo. */
COPER_API int coper_get_dims(const coper_handle* h, int64_t* F, int32_t* Ho, int32_t* Wo, int64_t* n_local);

/* Number of parameters the configuration requires and their leaf names / shapes, named as
 * the reference's TF variables: ent_emb, rel_emb, pred_bias, conv1_weights,
 * conv1_bias, fc_weights, fc_bias, "<name>/CPG/Projection<i>",
 * "<name>/CPG/Projection<i>/BatchNorm/{gamma,beta,moving_mean,moving_variance}",
 * "Conv1BN/...", "FCBN/...".  shape_out must hold 4 entries. */
COPER_API int coper_num_params(const coper_handle* h);
COPER_API int coper_param_spec(const coper_handle* h, int index, const char** leaf_name, int64_t* shape_out, int* ndim_out);

/* Replaces variable assignment / Saver.restore (models.py:203-336, run_cpg.py:205-206).
 * dev_ptr: fp32, contiguous, on cfg.device.  ent_emb / pred_bias are the LOCAL shard rows. */
COPER_API int coper_set_param(coper_handle* h, const char* leaf_name, const void* dev_ptr,
                    const int64_t* shape, int ndim);

/* coper_config.x3_ent_absmax after coper_create (a host that learns the table-wide maximum only when the parameters arrive:
 * coper_amd.models.ConvE.load_parameters, coper_amd/sharding.py).  The handle must be prepared (again) afterwards. */
COPER_API int coper_set_x3_ent_absmax(coper_handle* h, float absmax);

/* Builds everything inference derives from the parameters (re-run after any weight change):
 *   - BN (Conv1BN, FCBN, generator BNs) folded to per-channel affine   models.py:61-65,386-388,416-418
 *   - per-relation generated / looked-up conv filters, conv bias, dense weights, dense bias
 *     (ContextualParameterGenerator.generate / ParameterLookup.generate, models.py:56-76,90-94,
 *      338-352) evaluated once per relation id instead of once per sample
 *   - dense weights and the entity table re-laid out in MFMA-fragment-major order. */
COPER_API int coper_prepare(coper_handle* h, void* stream);

/* Pre-size the workspace so later calls with B <= max_queries and nnz <= max_filter_nnz do
 * not allocate (needed before hipGraph capture).  Optional: calls grow the workspace lazily. */
COPER_API int coper_reserve(coper_handle* h, int64_t max_queries, int64_t max_filter_nnz, void* stream);

/* Batch staging: the reference's placeholders hold int32 ids (e1, e2, rel: models.py:148-152; so do the indices a host
 * builds a CSR filter from), this ABI takes int64.  dst[i] = src[i] for n values; `src` may be device memory or PINNED,
 * device-mapped HOST memory (hipHostMalloc): the kernel then pulls the ids over PCIe itself -- one launch instead of a copy
 * engine transfer plus a widening pass.  No handle state is touched. */
COPER_API int coper_widen_ids(coper_handle* h, const int32_t* src, int64_t n, int64_t* dst, void* stream);

/* The same staging, overlapped: registers (src, n, dst) as a job the NEXT coper_encode / coper_encode_rank on this handle carries
 * out beside its encoder launch -- extra workgroups of that launch read the pinned batch over PCIe while the relation tiles
 * stream their weights (fewer tiles than CUs at the BASELINE shapes: they run on CUs that would idle), so the batch of pass
 * n + 1 arrives under pass n's kernels without a second stream.  dst must not be an array pass n itself reads (two staging
 * buffers, used alternately); configurations the fused encoder does not serve run the job as a launch of its own at the same
 * point.  Nothing is queued on a stream by this call; a later call replaces a job that has not run.  A pass that is being
 * captured into a hipGraph leaves the job pending for the next eager call (a replay must not repeat a read with the pointers
 * recorded at capture time). */
COPER_API int coper_stage_ids_next(coper_handle* h, const int32_t* src, int64_t n, int64_t* dst);

/* The way back: n int32 values (the ranks of a pass) from device memory to `dst`, which may be PINNED, device-mapped HOST
 * memory -- the kernel posts the writes over PCIe right behind the pass's last kernel (a copy-engine D2H on the same stream
 * starts ~12 us later on this runtime).  The host reads dst after synchronising with the stream. */
COPER_API int coper_copy_out_i32(coper_handle* h, const int32_t* src, int64_t n, int32_t* dst, void* stream);

/* The same copy, overlapped: registers (src, n, dst) as a job the NEXT call on this handle that groups a batch by relation
 * (coper_encode, coper_encode_rank, a training step of a parameter-lookup model) carries out beside its first launch, on that call's stream: the ranks of pass
 * n reach the host under pass n + 1's histogram instead of through a launch of their own behind pass n.  src must stay
 * unchanged until then (it is: the next pass writes its ranks in its third launch); the last pass of a loop is followed by
 * coper_copy_out_i32.  A later registration replaces a job that has not run; n == 0 cancels; a call that is being captured into
 * a hipGraph leaves the job to the next eager one.  Nothing is queued by this call. */
COPER_API int coper_post_i32_next(coper_handle* h, const int32_t* src, int64_t n, int32_t* dst);

/* The NEXT pass's grouping, overlapped.  Every pass starts by sorting its batch by relation (the reference instead materialises
 * per-query weights, models.py:350,412): two small launches that nothing else can overlap, 18 us of a 487 us pass at the BASELINE
 * shapes.  This call registers the id arrays of the pass that will FOLLOW the next one enqueued on this handle -- device addresses
 * (e1 / rel of that later coper_encode / coper_encode_rank call; have_e1_rows != 0: it will pass e1_rows, e1 is ignored), whose
 * contents are either there already or are what a coper_stage_ids_next job registered for the same launch will write there.  One
 * more workgroup of the next fused encoder launch then sorts that batch into a second set of grouping arrays while the relation
 * tiles stream their weights; the following coper_encode_rank call with exactly these pointers and B finds its grouping done and
 * starts with its encoder launch (a pending coper_post_i32_next job rides there instead).  Any other call in between (coper_encode
 * included: only a pass that writes ranks may run on a prepared grouping), a coper_prepare, a training step, a growing workspace
 * or a pass captured into a hipGraph drops the prepared grouping: the pass then groups itself as always.  The registration is for
 * the NEXT encode / encode_rank call only: a call that cannot carry it (captured into a hipGraph, a configuration the fused
 * encoder does not serve) drops it.  Nothing is queued by this call; B == 0 cancels.
 * THE GUARD.  The pointers identify the batch, not its contents: if the ids at those addresses are rewritten between the launch
 * that sorted them and the pass that consumes the sorting (the natural mistake with two staging buffers), that pass would encode
 * the ids that WERE there and rank them against the new e2 / filters.  So the consuming pass checks on the device: every tile
 * of its encoder launch compares the live (e1, rel) of each of its queries with what was sorted -- every query exactly once, its
 * loads beside the tile's own first loads -- and on ANY difference the pass's ranks are all written as COPER_RANK_STALE (negative:
 * no consumer can mistake them for ranks, which start at 1; n_equal / h of that call are undefined) and the pass is counted
 * (coper_stale_passes).  The caller ranks such a batch again with a plain call; coper_amd.metrics.ranking_and_hits and
 * bench.py do.  A stale grouping never reaches the ranks; a grouping that is not stale is never reported (tests/test_gpu_pipeline.py:
 * random interleavings of every registration, pass, prepare, training step, capture and workspace growth, ids rewritten at random). */
COPER_API int coper_group_next(coper_handle* h, const int64_t* e1, const int64_t* rel, int64_t B, int32_t have_e1_rows);

/* Row gather tf.nn.embedding_lookup(ent_emb, ids) (models.py:176) restricted to the shard:
 * out[b,:] = ent_emb[ids[b]] if shard_lo <= ids[b] < shard_hi else 0.  (Multi-GPU: sum over
 * ranks = the full gather.)  out: [B, d]. */
COPER_API int coper_gather_entities(coper_handle* h, const int64_t* ids, int64_t B, float* out, void* stream);

/* Replaces ConvE._create_predictions (models.py:354-426) in inference mode:
 * h_out[B, d] = predicted_e2_emb.  e1_rows: optional [B, d] pre-gathered ent_emb[e1] rows
 * (required when the shard does not hold every e1; NULL = gather locally). */
COPER_API int coper_encode(coper_handle* h, const int64_t* e1, const int64_t* rel, int64_t B,
                 const float* e1_rows, float* h_out, void* stream);

/* Replaces ConvE._compute_likelihoods(..., ent_indices=None) (models.py:428-437):
 * logits[B, n_local] = h . ent_emb_shard^T + pred_bias_shard  (no sigmoid). ld = row stride of logits. */
COPER_API int coper_score_all(coper_handle* h, const float* hvec, int64_t B, float* logits, int64_t ld, void* stream);

/* Replaces ConvE._compute_likelihoods(..., ent_indices=lookup) (models.py:438-443):
 * out[B, L] = h[b] . ent_emb[lookup[b,l]] + pred_bias[lookup[b,l]]; lookup holds GLOBAL ids,
 * entries outside the shard produce 0 (sum over ranks = full result). */
COPER_API int coper_score_lookup(coper_handle* h, const float* hvec, const int32_t* lookup, int64_t B, int64_t L,
                       float* out, void* stream);

/* Target scores, tgt: float [2 B].  tgt[b] = logit(b, e2[b]) in the mode's own arithmetic, bit-identical to the value
 * coper_score_all writes for that element (metrics.py:44); tgt[B + b] = the same logit by the fp32 chain of COPER_SCORE_F32
 * (equal to tgt[b] in that mode): what the x3 mode's exact band decides close comparisons against.  Entries whose e2 is not
 * in the shard are 0 in both halves: the sum over shards is the full result. */
COPER_API int coper_target_scores(coper_handle* h, const float* hvec, const int64_t* e2, int64_t B, float* tgt, void* stream);

/* out[b] = pred_bias_b + rows[b] . hvec[b] by the fp32 chain, for entity rows the caller holds (rows float [B, d], bias float
 * [B]): entity-sharded evaluation gathers ent_emb[e2] and pred_bias[e2] together with ent_emb[e1] in ONE all-reduce and
 * every rank then scores the targets from its copy -- no exchange of target logits (coper_amd/sharding.py; SURVEY 8(e)
 * step 1).  {out, out} is a valid `tgt` for coper_rank_counts. */
COPER_API int coper_score_rows(coper_handle* h, const float* hvec, const float* rows, const float* bias, int64_t B, float* out,
                               void* stream);

/* Replaces the host ranker loop of ranking_and_hits (metrics.py:44-50) without materialising
 * logits: for every query b over the shard's entities j, with the known-answer filter given
 * in CSR form (filt_idx[filt_indptr[b] .. filt_indptr[b+1]) = GLOBAL ids, sorted ascending;
 * the dense e2_multi mask of data.py:182-186 made sparse):
 *   n_greater[b] = #{ j != e2[b], j not filtered : logit(b,j) >  tgt[b] }
 *   n_equal[b]   = #{ j != e2[b], j not filtered : logit(b,j) == tgt[b] }
 * so that rank = 1 + sum_over_shards(n_greater) when tie-free (any value in
 * [1+n_greater, 1+n_greater+n_equal] under ties -- the reference's np.argsort is unstable).
 * n_equal may be NULL: ties are then not counted (the reference never computes them; saves one compare per
 * score) and rank = 1 + n_greater is the optimistic end of the band.
 * tgt: [2 B] global target scores (from coper_target_scores, summed over shards; or {t, t} from coper_score_rows): the count
 *   kernel of the x3 mode leaves every comparison closer than its own error (the band around tgt[b]) to the fp32 chain, which
 *   compares against tgt[B + b]: n_greater / n_equal of both modes are those of the fp32 chain on hvec.
 * filt_nnz = filt_indptr[B], passed from the host that built the CSR (sizes the launch).
 * k > 0 additionally returns the shard's top-k of the FILTERED row (target kept, like
 * metrics.py:46), order (score desc, id asc): topk_val [B,k] (-inf padded), topk_idx [B,k]
 * global ids (-1 padded).  k == 0: both may be NULL.  Not needed for ranks.  k <= 128 selects from
 * per-block maxima written by the count pass (no logits; workspace: one float per (32 entities, query) -- per (64 entities, query) on shards of 65,536 rows and more -- of a
 * query chunk + 32 floats per (k + filter entries)); otherwise logits are materialised chunk by chunk (<= 256 MiB). */
COPER_API int coper_rank_counts(coper_handle* h, const float* hvec, const float* tgt, const int64_t* e2,
                      const int64_t* filt_indptr, const int64_t* filt_idx, int64_t filt_nnz, int64_t B, int32_t k,
                      int32_t* n_greater, int32_t* n_equal, float* topk_val, int64_t* topk_idx,
                      void* stream);

/* Single-shard convenience: coper_target_scores + coper_rank_counts + rank = 1 + n_greater.
 * ranks: int32 [B]. */
COPER_API int coper_rank(coper_handle* h, const float* hvec, const int64_t* e2, const int64_t* filt_indptr,
               const int64_t* filt_idx, int64_t filt_nnz, int64_t B, int32_t* ranks, int32_t* n_equal,
               void* stream);

/* One evaluation batch end to end: coper_encode + coper_rank in one call -- what one `session.run` of the
 * reference's ranker loop computes (metrics.py:40-57).  h_out: optional float [B, d] (NULL: the embedding is
 * not needed; in the bf16x3 mode it then never exists in fp32 -- the dense finalize writes the operand planes
 * of the rank kernels directly).  Same results as the two calls, bit for bit.
 * Cost model of the filter: in the bf16x3 mode the workgroup that finalizes 32 consecutive queries also takes back their
 * known answers, the first 352 CSR entries of the block; entries beyond that (one (e1, rel) with 5,000 known tails) are
 * dealt over the whole chip by a second launch that costs ~2 us when no block needs it.  Callers have nothing to route. */
COPER_API int coper_encode_rank(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows,
                                const int64_t* e2, const int64_t* filt_indptr, const int64_t* filt_idx, int64_t filt_nnz,
                                int64_t B, float* h_out, int32_t* ranks, int32_t* n_equal, void* stream);

/* COPER_SCORE_BF16X3: the run-time audit of the exact band.  An AUDITED count launch (coper_config.band_audit_period: by default
 * the first after coper_prepare and every 8th, a sample of its workgroups, 128 pairs per round) re-scores pairs its band walk
 * decides (the competitors closest to each target) with the mode's own arithmetic as well and keeps the largest
 *        |logit_x3 - logit_chain| / (tau_q / 2)      (tau_q / 2: the error the band allows one logit; see rank_band_kappa)
 * seen since the last reset, and the number of pairs audited (modulo 2^32).  A ratio below 1 on every audited pair is what
 * makes the mode's ranks the fp32 chain's; the library's tests assert <= 0.5, coper_amd.metrics.ranking_and_hits logs a
 * warning above it.  Synchronises the stream.  COPER_SCORE_F32 handles report 0 / 0.  The reference has no counterpart: its
 * ranker works on materialised fp32 logits (metrics.py:40-50). */
COPER_API int coper_band_audit(coper_handle* h, int32_t reset, float* max_ratio, int64_t* n_pairs, void* stream);
/* The same two words without a synchronisation: dst2[0] = the ratio's float bits, dst2[1] = the pair count, posted to `dst2`
 * (device memory or PINNED, device-mapped host memory, like coper_copy_out_i32) behind the work already on the stream; a
 * host that waits for its ranks anyway reads them with the same wait.  COPER_ESTATE on a COPER_SCORE_F32 handle. */
COPER_API int coper_band_audit_post(coper_handle* h, int32_t reset, uint32_t* dst2, void* stream);
/* coper_copy_out_i32 and coper_band_audit_post in ONE launch: the n int32 ranks of a pass to dst[0, n) and the audit's two words to
 * dst[n], dst[n + 1] (zeros on a handle without a band), the audit reset for the next pass when reset != 0.  dst: n + 2 int32,
 * device or pinned host memory.  What `ranking_and_hits` fetches per `session.run` (metrics.py:40-43) plus the audit, behind the
 * pass's last kernel. */
COPER_API int coper_post_ranks_audit(coper_handle* h, const int32_t* ranks, int64_t n, int32_t* dst, int32_t reset, void* stream);
/* What a host DOES with the audit's two words (from coper_band_audit or coper_band_audit_post); host logic, no device work, no
 * synchronisation.  The contract of the path is integer ranks (metrics.py:44-50), so a measured error near the band's allowance
 * must not only be logged:
 *   ratio <= 0.5 (or no pair audited)  *action = COPER_BAND_KEEP;
 *   0.5 < ratio < 1                    the handle's kappa is doubled for every later pass         *action = COPER_BAND_WIDENED;
 *   ratio >= 1                         kappa is multiplied by the power of two that puts the error seen at or below a quarter of
 *                                      the new allowance, and the pass the words belong to must be ranked AGAIN by the caller
 *                                      (same call, same arguments: its close comparisons now go to the fp32 chain)
 *                                                                                                   *action = COPER_BAND_RERANK.
 * After a change the next count launch carries the audit whatever band_audit_period says, so a re-ranked pass is checked against
 * its new band at once.  The multiplier survives coper_prepare (it is a fact about the arithmetic, not about the weights);
 * *kappa_now = the relative half-width now in force.  coper_amd.metrics.ranking_and_hits applies this after every pass.
 * Sampling: with the default band_audit_period the first count launch after coper_prepare and every 8th from there are audited,
 * on the first round of at most ~512 workgroups and 128 pairs per round -- a sample; band_audit_period = 1 audits every launch.
 * No counterpart in the reference (it ranks materialised fp32 logits). */
#define COPER_BAND_KEEP 0
#define COPER_BAND_WIDENED 1
#define COPER_BAND_RERANK 2
COPER_API int coper_band_policy(coper_handle* h, float max_ratio, int64_t n_pairs, int32_t* action, float* kappa_now);

/* Timing hook used by bench.py: average device time (ms) of the dominant kernel
 * (score_count) over the launches since the last reset, measured with hipEvents recorded on
 * the launch stream.  enable != 0 turns per-launch event recording on. */
/* Step 1 of the entity-sharded exchange (coper_amd/sharding.py): every rank needs ent_emb[e1], ent_emb[e2] and pred_bias[e2] of the chunk
 * (models.py:176, :437); each row lives on one shard.  coper_pack_owned_rows writes what THIS shard owns for the all-gather --
 * buf[cap + 1][d + 1] float: row 0 = the header { hdr0, hdr1, 0... } (the shard's largest |ent_emb| and the table-wide maximum in
 * force: how the ranks keep ONE power of two for their entity planes), row 1 + i = { ent_emb[local_rows[i]], pred_bias[local_rows[i]] }
 * from the registered parameter tensors (no prepared state needed; a row number outside [0, n_local) gives a zero row and is counted
 * by coper_check_ids once the handle has a workspace), zeros up to cap -- and coper_unpack_rows hands the gathered rows
 * out: rows1[b] = gathered[take1[b]][0, d), rows2[b] = gathered[take2[b]][0, d), bias2[b] = gathered[take2[b]][d].  One launch each. */
COPER_API int coper_pack_owned_rows(coper_handle* h, const int64_t* local_rows, int64_t n, int64_t cap, float hdr0, float hdr1, float* buf,
                                    void* stream);
COPER_API int coper_unpack_rows(coper_handle* h, const float* gathered, const int64_t* take1, const int64_t* take2, int64_t B, float* rows1,
                                float* rows2, float* bias2, void* stream);

/* Step 3 of the entity-sharded exchange (coper_amd/sharding.py; north_star's one collective of counts + top-k): the record a shard
 * contributes, packed by ONE launch -- rec[B + 1][1 + 2 k] int64: row b = { n_greater << 32 | n_equal, the k top scores' float bits
 * (sign-extended int32), the k global ids } from coper_rank_counts' outputs; row B = the shard's band-audit words
 * { ratio bits << 32 | min(pairs, 2^31 - 1) } read on the device (and reset when reset_audit != 0: no host round trip; zeros on a
 * handle without a band) -- and, after the all-gather, the merge of all_rec[world][B + 1][1 + 2 k] by one more: ranks = 1 + the
 * summed n_greater (metrics.py:50), n_equal summed (may be NULL), the candidates side by side cand_val / cand_idx [B][world k]
 * (shard-major; the global top-k is their (score desc, id asc) selection).  Integer sums: bit-equal to the single-GPU ranks. */
COPER_API int coper_pack_shard_record(coper_handle* h, const int32_t* n_greater, const int32_t* n_equal, const float* topk_val,
                                      const int64_t* topk_idx, int64_t B, int32_t k, int32_t reset_audit, int64_t* rec, void* stream);
COPER_API int coper_merge_shard_records(coper_handle* h, const int64_t* all_rec, int32_t world, int64_t B, int32_t k, int32_t* ranks,
                                        int32_t* n_equal, float* cand_val, int64_t* cand_idx, void* stream);

/* Host only: n int64 ids narrowed to int32 into dst (the pinned staging buffer coper_widen_ids / coper_stage_ids_next read), checked on
 * the way: *status bit 0 = a value does not fit int32 (the caller then passes int64 arrays the ordinary way), bit 1 (with a CSR
 * indptr of n_rows rows over this array, indptr[0] = 0, indptr[n_rows] = n) = a row is not ascending (the filter contract of
 * coper_rank / coper_encode_rank: the caller sorts such rows first).  One pass instead of three over the bulk of a batch's host
 * marshalling (metrics.py:40-45 feeds placeholders; here the feed is the staging buffer). */
COPER_API int coper_pack_ids_i32(const int64_t* src, int64_t n, int32_t* dst, const int64_t* indptr, int64_t n_rows, int32_t* status);

/* Host only: the metrics of a pass's ranks as the reference computes them (metrics.py:53-57, 65-76): mean rank and mean reciprocal
 * rank as float64 means (the same summation order as numpy's np.mean, so the values are the ones the reference's expressions
 * give), hits[k] = the fraction of ranks <= levels[k].  ranks start at 1 (metrics.py:50); EINVAL otherwise and for n == 0. */
COPER_API int coper_hits_means(const int32_t* ranks, int64_t n, const int32_t* levels, int32_t n_levels, double* mean_rank, double* mrr,
                               double* hits);

/* One training batch of the reference's samplers (CoPER_ConvE/qa_cpg/data.py:228-311), built on the device by ONE launch (round 6).
 * Replaces what `train_dataset(...)`'s map function does per record on 32 tf.data threads (data.py:93-94, 138-156).  All pointers are
 * device memory on `device`:
 *   rec int64 [B]: the (e1, rel) record of each row of the batch (the host keeps the record stream, its shuffle buffer and the batching:
 *     data.py:136-160 on ids only); pos_tail int64 [B]: the row's positive (one positive per row: data.py:278-311; NULL when proportional);
 *   rec_e1, rec_rel int64 [n_rec]; tail_indptr int64 [n_rec + 1], tail_idx int64: the known train tails of every record (CSR);
 *   max_tails: the longest tail list among the batch's records (sizes the workgroups' hash set of known tails);
 *   proportional != 0: data.py:228-277 with prop_negatives -- the first `lead` of the record's tails in a fresh random order, then sampled
 *     entities; lead = the number of tails when that is <= int(L / (1 + prop_negatives)), L - min(num_ent, L - that) otherwise;
 * out: e1, rel, e2 int64 [B]; lookup int32 [B, L] (obj_lookup_values); labels float [B, L] (e2_multi: 1 where the looked-up entity is a
 *   known tail of the record -- a sampled "negative" that is one is supervised as positive, as the reference comments).
 * The sampled entities of a row are the head of a fresh uniform permutation of all entities (an ordered uniform sample without
 * replacement), a function of (seed, batch, row) alone.  L <= 2048, num_ent < 2^31 (the lookup's dtype); EUNSUPPORTED otherwise.
 * Asynchronous on `stream`; no handle: the sampler knows nothing of the model. */
COPER_API int coper_sample_train_batch(int32_t device, const int64_t* rec, const int64_t* pos_tail, const int64_t* rec_e1,
                                       const int64_t* rec_rel, const int64_t* tail_indptr, const int64_t* tail_idx, int64_t B, int64_t L,
                                       int64_t num_ent, int32_t proportional, double prop_negatives, int64_t max_tails, uint64_t seed,
                                       uint64_t batch, int64_t* e1, int64_t* rel, int64_t* e2, int32_t* lookup, float* labels, void* stream);

/* Ids are validated on the device and clamped, never trusted: returns in *n_bad the number of
 * out-of-range relation ids seen by coper_encode since the last call (synchronises the stream). */
COPER_API int coper_check_ids(coper_handle* h, int64_t* n_bad, void* stream);

/* The rank written for EVERY query of a coper_encode_rank pass that ran on a grouping prepared ahead (coper_group_next) whose id
 * arrays had been rewritten since: see "THE GUARD" there.  Whatever the count and band launches add to it stays negative. */
#define COPER_RANK_STALE (-(1 << 30))
/* Passes since the last call whose prepared grouping was found stale (synchronises the stream; the counter restarts at 0). */
COPER_API int coper_stale_passes(coper_handle* h, int64_t* n_passes, void* stream);

/* Device memory (bytes) currently held by all handles of this process: every allocation of the library is entered in a
 * ledger.  Returns to its previous value when a handle is destroyed (tests/test_gpu_train.py checks exactly that).  The
 * reference has no counterpart: TensorFlow owns its allocations (`tf.Session` of run_cpg.py:111). */
COPER_API int64_t coper_live_device_bytes(void);

COPER_API int coper_profile_enable(coper_handle* h, int enable);
COPER_API int coper_profile_read(coper_handle* h, const char* kernel, double* total_ms, int64_t* launches);

/* ---------------------------------------------------------------------------------------------
 * Training step: replaces `session.run(model.train_op)` of run_cpg.py:211-219, i.e.
 * models.py:176-200 (train-mode forward, sampled scorer, label-smoothed sigmoid cross-entropy),
 * tf.clip_by_global_norm(5.0) (models.py:199) and utils/amsgrad.py:130-189.
 * Supported: static (plain ConvE), g_linear / g_MLP generated or g_lookup dense layer; static, generated or
 * looked-up conv filters; concat_rel.  Looked-up conv filters with a static dense layer: COPER_EUNSUPPORTED.
 * The parameters registered with coper_set_param are UPDATED IN PLACE (they are the variables), including
 * the BN moving statistics; the caches built by coper_prepare go stale, so the handle must be prepared
 * again before the next inference call (enforced).
 * Dropout uses a counter-based hash of (seed, step, stage, element) instead of TF's RNG stream.
 * ------------------------------------------------------------------------------------------- */
typedef struct coper_train_config {
  int32_t abi_version;            /* COPER_ABI_VERSION */
  float learning_rate;            /* model_descriptors['learning_rate'] (models.py:130) */
  float beta1, beta2, epsilon;    /* AMSGrad defaults 0.9, 0.999, 1e-8 (amsgrad.py:60-62) */
  float clip_norm;                /* 5.0 (models.py:199) */
  float label_smoothing_epsilon;  /* models.py:101,450 */
  float hidden_dropout;           /* after conv BN ReLU (models.py:390) */
  float output_dropout;           /* before FCBN (models.py:414) */
  float batch_norm_momentum;      /* moving-average DECAY (models.py:64,387,417) */
  int32_t batch_norm_train_stats; /* BN uses batch statistics while training (models.py:62,358) */
  uint32_t seed;                  /* dropout stream */
  float context_rel_dropout;      /* dropout inside the g_MLP generators (models.py:67-68,118) */
  int32_t reserved[7];
} coper_train_config;

/* Allocates gradients and the AMSGrad slots m, v, v_hat (zeros) for every trainable parameter; every
 * parameter must have been registered with coper_set_param. */
COPER_API int coper_train_init(coper_handle* h, const coper_train_config* cfg);
/* One optimisation step on a training batch in the reference batch contract (models.py:139-152):
 * e1, rel int64 [B]; lookup int32 [B,L] (obj_lookup_values); labels float [B,L] (e2_multi for the looked-up
 * entities).  lookup == NULL with L == num_ent is 1-vs-all training (use_negative_sampling = False, run_cpg.py:116:
 * labels are the dense e2_multi [B, num_ent], models.py:159-162).  loss_out: device float[1], the batch loss
 * (models.py:448-453).
 * The registered tensors ARE the variables and the step keeps facts about them from one step to the next (round 6: the largest
 * magnitude of the dense weights, noted by the optimizer's pass as it writes them, is what the next step's operand packing scales
 * by).  A caller that changes parameter CONTENTS between steps by other means than this call registers the tensor again
 * (coper_set_param on the same pointer is enough; coper_amd.ConvE.load_parameters does) -- as coper_prepare asks for inference.
 * Ordering: to the caller the step is one sequence of launches on `stream`.  Inside, two stretches that do not depend on the chain
 * beside them (the scorer's backward; the projection gradient's product) run on streams of the training state's own, forked from
 * `stream` and joined to it by events before the call returns -- nothing of the step is left outside `stream`'s order. */
COPER_API int coper_train_step(coper_handle* h, const int64_t* e1, const int64_t* rel, const int32_t* lookup,
                               const float* labels, int64_t B, int64_t L, float* loss_out, void* stream);
/* The train-mode graph WITHOUT the update: what `session.run(model.loss)` or `session.run(model.predictions_lookup)` under
 * `{model.is_train: True}` evaluate when `train_op` is not among the fetches (models.py:183-192: the loss and the likelihoods are
 * plain tensors of the graph; the UPDATE_OPS and the optimizer hang off train_op only, models.py:194-200).  Same batch contract as
 * coper_train_step; dropout with the masks the NEXT coper_train_step will draw (the step counter is not advanced), batch
 * statistics when batch_norm_train_stats -- and nothing written: no variable, no BN moving statistic, no optimizer slot.
 * loss_out: device float[1] or NULL; pred_out: device float [B, L] logits (predictions_lookup; [B, num_ent] with lookup == NULL)
 * or NULL; h_out: device float [B, d] (predicted_e2_emb in training mode) or NULL.  Inference caches stay valid. */
COPER_API int coper_train_forward(coper_handle* h, const int64_t* e1, const int64_t* rel, const int32_t* lookup,
                                  const float* labels, int64_t B, int64_t L, float* loss_out, float* pred_out, float* h_out,
                                  void* stream);
/* Diagnostics: copies the (unclipped) gradient of the last step for a trainable leaf into `out` (device float
 * buffer of `cap` elements; may be NULL), returns its length in *n, and in *global_norm (optional, host) the
 * global gradient norm of the last step (synchronises).  Call it before anything regroups a batch by relation (an evaluation pass): the
 * looked-up dense table's rows of relations the last batch did not hold are not written by a step -- optimizer and norm skip them by the
 * batch's relation counts -- and are handed out as the zeros they stand for by those same counts. */
COPER_API int coper_train_grad(coper_handle* h, const char* leaf_name, float* out, int64_t cap, int64_t* n,
                               double* global_norm, void* stream);

/* Optimizer state, for checkpoints (the reference's tf.train.Saver stores the slots `<var>/AMSGrad`, `/AMSGrad_1`,
 * `/AMSGrad_2` = m, v, v_hat of utils/amsgrad.py:117-119 and the non-slot variables beta1_power / beta2_power,
 * amsgrad.py:108-113).  `which`: 0 m, 1 v, 2 v_hat; `buf`: device float buffer of `cap` elements (may be NULL to
 * query *n); set == 0 copies slot -> buf, set != 0 copies buf -> slot, stream-ordered.  Powers: get with non-NULL
 * outputs, set with non-NULL inputs; `step` is the dropout counter of the next step. */
COPER_API int coper_train_slot(coper_handle* h, const char* leaf_name, int32_t which, float* buf, int64_t cap, int32_t set,
                               int64_t* n, void* stream);
COPER_API int coper_train_powers(coper_handle* h, const double* set_beta1_power, const double* set_beta2_power,
                                 const int64_t* set_step, double* beta1_power, double* beta2_power, int64_t* step);

#ifdef __cplusplus
}
#endif
#endif /* COPER_HIP_H_ */
