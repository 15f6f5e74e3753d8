// Internal declarations shared by the libcoper_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/coper_hip.h"

// ---- build switches (round 6: pruned) -------------------------------------------------------------------------------------------
// The kernels keep ONE family of compile-time switches: the ablation / clock-stamp set COPER_DBG_* that the measurements of
// DESIGN_LOG.md were made with (a stage removed, a clock stamped per workgroup).  They exist in DIAGNOSTIC builds only: without
// -DCOPER_DIAG every one of them is undefined here, whatever the command line says, so the product library is one program.
// (tools/ab_build.py passes -DCOPER_DIAG with them.)  The variant builds that rounds 2 - 5 measured and rejected -- the bf16 split,
// the fp32-FMA conv of the fused encoder, loads inside the count kernel's asm blocks, buffer loads, the 8-bit second term, the
// old grid layouts ... -- are gone from the sources; their numbers are in DESIGN_LOG.md.  What remains besides are named tunables
// with ONE shipped value (`#ifndef X / #define X value`: COPER_SC3_MB, COPER_TL_WAVES, COPER_FUSED_PBUDGET, ...).
#ifndef COPER_DIAG
#undef COPER_DBG_AMSGRAD_NO_NT
#undef COPER_DBG_CLOCK
#undef COPER_DBG_DENSE_NO_DMA
#undef COPER_DBG_DENSE_NO_MFMA
#undef COPER_DBG_FUSED_CLOCK
#undef COPER_DBG_FUSED_EXIT
#undef COPER_DBG_FUSED_NO_CONV
#undef COPER_DBG_FUSED_NO_IMG
#undef COPER_DBG_FUSED_NO_MFMA
#undef COPER_DBG_FUSED_NO_STORE
#undef COPER_DBG_FUSED_NO_W
#undef COPER_DBG_FUSED_NO_XREAD
#undef COPER_DBG_GROUP_CLK
#undef COPER_DBG_NO_EPILOGUE
#undef COPER_DBG_NO_GLOADS
#undef COPER_DBG_NO_GROUP_CHK
#undef COPER_DBG_NO_LDS
#undef COPER_DBG_REG_NO_BAR
#undef COPER_DBG_REG_NO_MFMA
#undef COPER_DBG_REG_NO_X
#undef COPER_DBG_SC3_DUMMY_LDS
#undef COPER_DBG_SC3_EPI_R0
#undef COPER_DBG_SC3_HALF_LDS
#undef COPER_DBG_SC3_LOADS_FIRST
#undef COPER_DBG_SC3_NO_BAND
#undef COPER_DBG_SC3_NO_EPI
#undef COPER_DBG_SC3_ONE_LDS
#undef COPER_DBG_SC3_SKIP_GL
#undef COPER_DBG_SC3_SKIP_LDS
#undef COPER_DBG_TK_OVER
#undef COPER_DBG_TL_CLOCK
#undef COPER_DBG_TL_NOGATHER
#undef COPER_DBG_W128_NO_EPI
#undef COPER_DBG_W128_NO_LOADS
#endif

namespace coper {

// every device allocation of the library goes through these two: a process-wide ledger (pointer -> bytes) behind
// coper_live_device_bytes(), so that a leak shows up as a number and not as a guess from hipMemGetInfo (which also moves with
// the runtime's own scratch and pool decisions)
hipError_t tracked_malloc_impl(void** p, size_t bytes);
hipError_t tracked_free(void* p);
template <typename T>
inline hipError_t tracked_malloc(T** p, size_t bytes) { return tracked_malloc_impl((void**)p, bytes); }


struct Param {
  const float* ptr = nullptr;
  std::vector<int64_t> shape;
  bool set = false;
};

struct ParamSpec {
  std::string name;
  std::vector<int64_t> shape;
};

// Derived sizes, exactly as ConvE._create_variables derives them (models.py:261-271).
struct Dims {
  int64_t E = 0, R = 0;
  int d = 0, r = 0, emb_h = 0, emb_w = 0;
  int fh = 3, fw = 3, C = 32;
  bool concat_rel = false, lookup = false, gen_conv = false, gen_fc = false, stacked = false, ctx_bn = false;
  int in_h = 0, in_w = 0, Ho = 0, Wo = 0;
  int64_t F_conv = 0, F = 0;
  // padded sizes used by the MFMA layouts
  int64_t F_pad = 0;   // F rounded up to 32 (two 16x16x4 f32 super-steps = one 16x16x32 bf16 k-step)
  int d_pad16 = 0;     // d rounded up to 16 (feature blocks of the dense layer)
  int nfb = 0;         // d_pad16 / 16
  int d_pad8 = 0;      // d rounded up to 8 (k-steps of the score kernels)
  int KS = 0;          // d_pad8 / 8
  int KS16 = 0;        // ceil(d / 16): k-steps of the bf16 MFMA
  int64_t n_local = 0; // shard rows
  int64_t n_eblk = 0;  // 32-row entity blocks (padded to a multiple of EBLK_ALIGN)
};

// half-width of the exact band of the x3 ranker, relative to |h_q| max|E_e| + max|bias| (per logit; a comparison of two
// logits gets twice that).  The largest error of the mode's logits against the fp32 chain measured over 3e8 logits of the
// FB15k-237-shaped pass (tools/rank_decomp.py): fp16 split 1.8e-7 |h_q||E_e| (rms 1.2e-8) -- the fp32 chain itself is
// 2.3e-7 from float64 --; round 2's bf16 split: 3.3e-6 (rms 3.1e-7).  The default keeps a factor 5 above
// the worst case seen, 80 (30) sigma.  coper_config.rank_band_kappa overrides them; the proven worst case of the split and
// of fp32 accumulation in any order is 3 * 2^-22 (2^-16) + 2 (3 * 16 KS16 + 1) 2^-24 ~ 7.5e-5 (1.2e-4) at d = 200.
constexpr float COPER_BAND_KAPPA_DEFAULT = 1e-6f;
constexpr int COPER_TOPK_PRUNED_MAX = 128;   // largest k served by the block-maxima top-k (bf16x3); above: logits chunks
constexpr int BAND_NCONST = 8;
// the check words of a grouping prepared ahead (group_body.h: the guard), 64-bit each
enum { GROUP_CHK_REL = 0, GROUP_CHK_E1 = 1, GROUP_CHK_PERM = 2, GROUP_CHK_LO = 3, GROUP_CHK_NLOCAL = 4, GROUP_CHK_RALL = 5,
       GROUP_CHK_STALE = 6, GROUP_CHK_WORDS = 8 };
constexpr int EBLK_ALIGN = 16;  // entity blocks consumed per workgroup iteration in score_count (8 waves x 2)

struct Timer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double total_ms = 0.0;
  int64_t launches = 0;
};

}  // namespace coper

struct coper_handle {
  coper_config cfg;
  coper::Dims dm;
  std::vector<coper::ParamSpec> specs;
  std::map<std::string, coper::Param> params;
  std::string err;
  bool prepared = false;

  // ---- derived device buffers (owned) ----
  float* conv_scale = nullptr;  // [C]   folded Conv1BN
  float* conv_shift = nullptr;
  float* fc_scale = nullptr;    // [d]   folded FCBN
  float* fc_shift = nullptr;
  float* conv_w_rel = nullptr;  // [R, fh*fw, C] generated / looked-up conv filters (gen_conv)
  float* conv_b_rel = nullptr;  // [R, C]
  float* fc_b_rel = nullptr;    // [R, d]  (gen_fc)
  float* Wf = nullptr;          // dense weights, fragment-major: [Rw][nfb][F_pad/16][64] float4
  int64_t Rw = 0;               // R when gen_fc else 1
  float* Ef = nullptr;          // entity table, fragment-major: [n_eblk][KS][64] float4
  float* bias_pad = nullptr;    // [n_eblk*32], -inf padded
  void* Wf16_hi = nullptr;      // COPER_SCORE_BF16X3: dense weights as hi / lo bf16 planes (16x16x32 fragment order)
  void* Wf16_lo = nullptr;
  bool enc_bf16 = false;        // encoder runs in bf16x3 (needs 3x3 filters, C % 8 == 0)
  void* Ef16_hi = nullptr;      // COPER_SCORE_BF16X3: entity table hi / lo bf16 planes, fragment-major
  void* Ef16_lo = nullptr;      //   [n_eblk][KS16][64] x 16 B
  void* Erm16_hi = nullptr;     //   row-major twins [n_eblk*32][KS16*16] bf16 (pair kernel gathers)
  void* Erm16_lo = nullptr;
  void* Ef3 = nullptr;          //   the count kernel's image (bf16x3_chain.h): [2 n_eblk][NS][2][64] x 16 B, zero-filled first
  unsigned* band_consts = nullptr;   // [BAND_NCONST] float bits: [0] max |E_e|_2, [1] max |pred_bias| of the shard (exact band), [2] max |E element|,
                                     //   [3] the band audit's largest |x3 - chain| / (tau / 2), [4] its pair count (coper_band_audit)
  float band_kappa_mult = 1.f;       // coper_band_policy: a power of two on top of the configured kappa (kept across coper_prepare)
  unsigned band_launches = 0;        // count launches since prepare (which of them the band audit rides on: kernels_score3_bf16.hip)
  int img_exp = 0;                   // e_I: the fused encoder's image planes (compute_x_exp)
  int x3_ent_exp = 0;                // e_E: the entity planes hold E 2^e_E (split16.h; prepare)
  float x3_ent_absmax = 0.f;         //   the maximum it was chosen from (the shard's, or coper_config.x3_ent_absmax)
  // coper_group_next: grouping arrays in sets.  Set 0 is HOME (the arrays below: every pass that groups itself, and every captured
  // graph, uses it); sets 1 and 2 are written by the grouping role of a fused encoder launch for the NEXT pass while the running
  // pass reads its own.  The fields perm / inv_perm / ... / x3m / fused_fin_dev always name the set of the pass being enqueued.
  struct GroupSet {
    int32_t* slab = nullptr;         // sets 1, 2: one allocation; set 0: unused (a snapshot of the home pointers)
    int32_t* rel_count = nullptr; int32_t* rel_offset = nullptr; int32_t* perm = nullptr; int32_t* inv_perm = nullptr;
    int32_t* sorted_row = nullptr; int32_t* sorted_rid = nullptr; int32_t* tiles = nullptr; int32_t* n_tiles = nullptr;
    float* x3m = nullptr; void* fused_fin_dev = nullptr; const int32_t* fused_fin_perm = nullptr;
    int64_t* chk = nullptr;          // sets 1, 2: [GROUP_CHK_WORDS] what the grouping role sorted (group_body.h: the guard of a prepared grouping)
  } gset[3];
  int gcur = 0;                      // the set the fields name now
  // The jobs a host registers for a LATER launch of this handle, and the grouping a launch prepared ahead: ONE owner, one
  // invalidation rule (the drop rules used to be spread over five places).
  //   stage  (coper_stage_ids_next): a batch to bring in beside the next fused encoder launch            n > 0: pending
  //   post   (coper_post_i32_next):  int32 results to copy out beside the next pass's first launch       n > 0: pending
  //   gnext  (coper_group_next):     id arrays of the pass AFTER the next one, to be sorted inside the next fused encoder launch
  //   gdone: a grouping launched ahead into set `set`; the pass that comes with exactly these pointers consumes it -- and CHECKS it
  //          on the device against the live ids (pass_chk below)
  // Lifetimes: stage / post name buffers the caller owns and stay pending until a launch carries them (an eager pass, or post_flush /
  // a launch of their own where no launch can carry them); gnext is for the NEXT encode / encode_rank call only; gdone is for the
  // call after that only.  invalidate_grouping() -- coper_prepare, a training step, a growing workspace, freed grouping sets -- drops
  // gnext and gdone together; nothing else does.
  struct PassPipeline {
    struct { const int32_t* src = nullptr; int64_t n = 0; int64_t* dst = nullptr; } stage;
    struct { const int32_t* src = nullptr; int64_t n = 0; int32_t* dst = nullptr;
             bool here = false; } post;  // here: the pass being enqueued was grouped ahead, its fused encoder launch carries the job
    struct { const int64_t* e1 = nullptr; const int64_t* rel = nullptr; int64_t B = 0; int rows = 0;
             bool pending = false;       // registered, not yet launched
             bool ride = false; } gnext; // the fused encoder launch being enqueued carries it (sets 1, 2 are allocated)
    struct { const int64_t* e1 = nullptr; const int64_t* rel = nullptr; int64_t B = 0; int rows = 0;
             bool done = false; int set = 0; } gdone;
    void invalidate_grouping() { gnext.pending = false; gnext.ride = false; gdone.done = false; }
    // the next call's view of a grouping prepared ahead: consumed (true) or dropped -- never kept for a later call
    bool take_prepared(const int64_t* e1, const int64_t* rel, int64_t B, int rows, int* set) {
      const bool hit = gdone.done && gdone.rel == rel && gdone.e1 == (rows ? nullptr : e1) && gdone.B == B && gdone.rows == rows;
      gdone.done = false;
      *set = hit ? gdone.set : 0;
      return hit;
    }
    int64_t take_stage() { const int64_t n = stage.n; stage.n = 0; return n; }
    int64_t take_post() { const int64_t n = post.n; post.n = 0; return n; }
  } pipe;
  // the pass being enqueued runs on a grouping prepared ahead: the set's check words (its tiles compare the live ids with what was
  // sorted; the kernels that preset the rank counters read the verdict); nullptr: the pass grouped itself
  int64_t* pass_chk = nullptr;
  int64_t stale_passes_host = 0;     // coper_stale_passes: what the device counter (group_done + 2) held when a growing workspace replaced it
  void* fused_fin_dev = nullptr;     // FusedFinConst (kernels_dense_fused_bf16.hip): the fused encoder's finalize constants, in device memory
  const int32_t* fused_fin_perm = nullptr;   //   the workspace generation they were written for
  int32_t* w_exp = nullptr;          // [R or 1] e_W per relation id: the dense-weight planes hold W_r 2^e_W (split16.h; prepare)
  int w_div = 1, w_rem = 0;          // coper_config.rel_mod_*: the weight planes hold the relations r with r % w_div == w_rem, relation r at slot r / w_div
  int x_exp = 0;                     // e_x: the conv activations enter the dense layer as x 2^e_x (from a bound; prepare)
  int32_t* x3s = nullptr;            // [4] the packed batch's exponents: [0] e_h, [1] e_E + e_h (bf16x3_chain.h)
  float* x3m = nullptr;              // [X3M_SLOTS] per-block maxima of the h rows being packed (bf16x3_chain.h)
  float* ctx_tmp[2] = {nullptr, nullptr};  // generator hidden activations
  size_t ctx_tmp_elems = 0;

  // ---- workspace (owned, grown lazily) ----
  int64_t ws_queries = 0;
  int64_t ws_nnz = 0;
  int ws_ksplit = 0;
  int32_t* rel_count = nullptr;   // [R+2] counts of the LAST grouping call (+ the out-of-range counter at R+1) = rel_count_buf[0]
  int32_t* rel_count_buf[2] = {nullptr, nullptr};   // [0] the published counts; [1] the accumulation buffer of k_rel_hist_scan (zero between calls)
  int32_t* group_done = nullptr;  // ticket counter of the histogram launch (its last block runs the scan)
  int32_t* rel_offset = nullptr;  // [R+1]
  int32_t* rel_cursor = nullptr;  // [R]
  int32_t* perm = nullptr;        // [B] sorted position -> query
  int32_t* inv_perm = nullptr;    // [B] query -> sorted position
  int32_t* sorted_row = nullptr;  // [B] sorted position -> local entity row of e1 (or the e1_rows row), -1 = not on this shard
  int32_t* sorted_rid = nullptr;  // [B] sorted position -> validated relation id
  int32_t* tiles = nullptr;       // [T_max*4] (rel, start, n, pad)
  int32_t* n_tiles = nullptr;     // [2] #small tiles, #big 16-query blocks
  int32_t* blk_off = nullptr;     // [R+1] exclusive scan of the big groups' block counts
  float* x_sorted = nullptr;      // [B, F_pad]
  float* z_part = nullptr;        // [ksplit, B, d_pad16]
  float* tgt_ws = nullptr;        // [B]
  int32_t* cnt_ws = nullptr;      // [2B]
  float* h_ws = nullptr;          // internal h of coper_encode_rank when the caller does not want it (fp32 mode)
  int64_t h_ws_rows = 0;
  float* logits_ws = nullptr;     // top-k path only: [chunk_rows, n_local]
  int64_t logits_ws_rows = 0;
  // pruned top-k (k <= COPER_TOPK_PRUNED_MAX; kernels_topk_bf16.hip)
  int64_t gmax_max_floats = (int64_t)1 << 28;   // set from the device memory size at prepare
  float* gmax_ws = nullptr;       // [n_eblk][query chunk]: block maxima written by the count pass
  size_t gmax_cap = 0;
  int32_t* cand_blk_ws = nullptr; // [k*B + nnz] candidate blocks, query q's at k*q + indptr[q]
  float* cand_val_ws = nullptr;   // [k*B + nnz][32] their logits
  size_t cand_cap = 0;
  int32_t* cand_q_ws = nullptr;   // [k*B + nnz] the query of every candidate slot
  char* tk_coarse_ws = nullptr;   // the threshold kernel's coarse level (topk_coarse_bytes)
  size_t tk_coarse_cap = 0;
  uint32_t* cand_tau_ws = nullptr; // [B] selection threshold per query (ordered float bits; 0: none)
  size_t cand_tau_cap = 0;
  int32_t* cand_sorted_ws = nullptr;  // candidate slots grouped by entity block, 32-padded per block
  int32_t* blk_cnt_ws = nullptr;  // [2 n_eblk] slots per block | scatter cursors
  int32_t* blk_off_ws = nullptr;  // [n_eblk + 1] (+ the scan's chunk sums)
  void* hfrag16_hi = nullptr;     // bf16x3: h hi / lo planes in fragment order
  void* hfrag16_lo = nullptr;
  void* hrm16_hi = nullptr;       //   row-major twins
  void* hrm16_lo = nullptr;
  void* hf3_ws = nullptr;         //   the count kernel's query image [ceil(B/128) * 8][NS][2][64] x 16 B (zero-filled at allocation)
  float* tband_ws = nullptr;      // [B] float2 {t_lo, t_hi}: the exact band around the mode's target logit
  float* tgtx_ws = nullptr;       // [B] exact-chain targets of the two-call flows
  void* mask_ws = nullptr;        // band bits of one count launch (kernels_score3_bf16.hip)
  size_t mask_cap = 0;            //   bytes
  const float* packed_hvec = nullptr;  // what hfrag16 currently holds (only trusted inside coper_rank)
  int64_t packed_B = 0;
  bool trust_packed = false;
  int32_t* row_of_ws = nullptr;   // bf16x3: CSR entry -> query row [nnz]
  bool excess_pending = false;    // the next band launch also runs the excess role (kernels_score3_bf16.hip), with
  alignas(8) unsigned char excess_args[128] = {};   //   these FilterArgs
  int32_t* heavy_ws = nullptr;    // fused tail: [0] number of listed blocks, [1] finished workgroups, [2..] 32-query blocks whose CSR entries
                                  //   exceed what their workgroup corrects itself (k_filter_excess_bf16x3); zero between passes
  int64_t row_of_cap = 0;
  float* hfrag_ws = nullptr;      // h re-packed in MFMA-fragment order [ceil(B/128)*4][KS][64] float4
  int num_cus = 256;
  void* train = nullptr;          // coper::TrainState (coper_train.hip)
  // coper_rank only: the counters accumulate straight into `ranks` (base 1) and are preset by the packing launch
  int32_t count_base = 0;
  int32_t* preset_cnt = nullptr;  // request: the next pack launch presets these [B] counters (and preset_eq to 0)
  int32_t* preset_eq = nullptr;
  const int64_t* expand_indptr = nullptr;      // request: the next target pass also writes row_of_ws for this CSR
  const int64_t* rows_expanded_for = nullptr;  // done: row_of_ws holds the row ids of this CSR
  const int32_t* counts_preset = nullptr;  // done: the next score_count on this buffer skips its own zeroing
  bool dense_attr_done = false;
  bool fused_attr_done = false;
  bool dense_small_only = false;  // set per launch: tiles above 32 queries go to the fused conv + dense kernel

  bool profile = false;
  std::map<std::string, coper::Timer> timers;
  std::vector<hipEvent_t> event_pool;   // recycled by coper_profile_read
};

namespace coper {

int fail(coper_handle* h, int code, const std::string& msg);
void train_destroy(coper_handle* h);   // coper_train.hip
void train_params_changed(coper_handle* h);
int hip_fail(coper_handle* h, hipError_t e, const char* what);

#define COPER_HIP_TRY(h, expr)                                    \
  do {                                                            \
    hipError_t _e = (expr);                                       \
    if (_e != hipSuccess) return coper::hip_fail((h), _e, #expr); \
  } while (0)

// kernels_prepare.hip
int launch_fold_bn(coper_handle* h, const float* gamma, const float* beta, const float* mean, const float* var,
                   int n, float eps, float* scale, float* shift, hipStream_t s);
// out[rel, n] = act( sum_k ctx[rel,k] * P[k,n] ), optional BN+ReLU epilogue (generator hidden layers,
// conv filters/biases, dense bias): small N.
int launch_gen_small(coper_handle* h, const float* ctx, int64_t R, int K, const float* P, int64_t N,
                     const float* bn_scale, const float* bn_shift, bool relu, float* out, hipStream_t s);
// dense weights into fragment-major Wf.  mode 0: sum_k ctx[rel,k]*P[k, f*d+i]; mode 1: P[rel, f*d+i] (lookup
// and static: ctx == nullptr).
int launch_gen_dense_frag(coper_handle* h, const float* ctx, int64_t R, int K, const float* P, int mode,
                          float* Wf, hipStream_t s);
int launch_entity_frag(coper_handle* h, const float* ent, const float* bias, hipStream_t s);

// kernels_encode.hip
int launch_group_by_relation(coper_handle* h, const int64_t* e1, const int64_t* rel, bool have_e1_rows, int64_t B, int tq,
                             hipStream_t s);
int launch_conv(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, int64_t B,
                hipStream_t s);
int launch_dense(coper_handle* h, const int64_t* rel, int64_t B, int tq, int ksplit, float* h_out, hipStream_t s);
int launch_copy_i32(coper_handle* h, const int32_t* src, int64_t n, int32_t* dst, hipStream_t s);
int launch_pack_owned_rows(coper_handle* h, const float* ent, const float* bias, const int64_t* loc, int64_t n, int64_t cap, float hdr0, float hdr1,
                           float* buf, hipStream_t s);
int launch_unpack_rows(coper_handle* h, const float* all, const int64_t* take1, const int64_t* take2, int64_t B, float* g1, float* g2, float* b2,
                       hipStream_t s);
int launch_pack_shard_record(coper_handle* h, const int32_t* ng, const int32_t* ne, const float* tv, const int64_t* ti, int64_t B, int k,
                             unsigned* audit, int reset, int64_t* rec, hipStream_t s);
int launch_merge_shard_records(coper_handle* h, const int64_t* all, int world, int64_t B, int k, int32_t* ranks, int32_t* ne, float* vals,
                               int64_t* ids, hipStream_t s);
int launch_copy_i32_audit(coper_handle* h, const int32_t* src, int64_t n, int32_t* dst, unsigned* audit, int reset, hipStream_t s);
int launch_widen_ids(coper_handle* h, const int32_t* src, int64_t n, int64_t* dst, hipStream_t s);
int launch_gather_entities(coper_handle* h, const int64_t* ids, int64_t B, float* out, hipStream_t s);

// kernels_score.hip
int launch_score_all(coper_handle* h, const float* hvec, int64_t B, float* logits, int64_t ld, hipStream_t s);
int launch_score_count(coper_handle* h, const float* hvec, const float* tgt, int64_t B, int32_t* ng, int32_t* ne,
                       hipStream_t s);
int launch_pair_targets(coper_handle* h, const float* hvec, const int64_t* e2, int64_t B, float* tgt, hipStream_t s);
int launch_filter_correct(coper_handle* h, const float* hvec, const float* tgt, const int64_t* e2,
                          const int64_t* indptr, const int64_t* idx, int64_t nnz, int64_t B, int32_t* ng,
                          int32_t* ne, hipStream_t s);
int launch_score_lookup(coper_handle* h, const float* hvec, const int32_t* lookup, int64_t B, int64_t L, float* out,
                        hipStream_t s);
int launch_finish_ranks(coper_handle* h, const int32_t* ng, int64_t B, int32_t* ranks, hipStream_t s);
int score_kernels_init(coper_handle* h);
// kernels_score_bf16.hip (COPER_SCORE_BF16X3)
int launch_rows_to_frag_bf16(coper_handle* h, const float* src, int64_t n_rows, int64_t n_blk, uint4* hi, uint4* lo,
                             uint4* rm_hi, uint4* rm_lo, uint4* f3, bool query_side, hipStream_t s);
int launch_pack_h_bf16(coper_handle* h, const float* hvec, int64_t B, hipStream_t s);
int launch_absmax_publish(coper_handle* h, const float* src, int64_t n, hipStream_t s);
int launch_score_count_bf16x3(coper_handle* h, const float* hvec, const float* tgt_x, const int64_t* e2, const int64_t* indptr,
                              const int64_t* idx, int64_t B, int32_t* ng, int32_t* ne, hipStream_t s);
int launch_score_all_bf16x3(coper_handle* h, const float* hvec, int64_t B, float* logits, int64_t ld, hipStream_t s);
int launch_pair_targets_bf16x3(coper_handle* h, const float* hvec, const int64_t* e2, int64_t B, float* tgt,
                               hipStream_t s);
int launch_score_lookup_bf16x3(coper_handle* h, const float* hvec, const int32_t* lookup, int64_t B, int64_t L,
                               float* out, hipStream_t s);
int launch_filter_correct_bf16x3(coper_handle* h, const int64_t* e2, const int64_t* indptr, const int64_t* idx, int64_t nnz,
                                 int64_t B, int32_t* ng, hipStream_t s);
int score_bf16_kernels_init(coper_handle* h);
// queries whose block maxima are held at a time: at most `cap_floats` (1/32 of the device memory, at least 1 GiB;
// the threshold kernel runs one workgroup per 16 or 32 queries, so short chunks leave the chip idle) or one tile
inline int64_t topk_chunk_queries(int64_t n_eblk, int64_t B, int64_t cap_floats) {
  int64_t qc = cap_floats / n_eblk / 128 * 128;
  if (const char* e = getenv("COPER_TOPK_CHUNK_QUERIES")) qc = atoll(e) / 128 * 128;   // tests: force several chunks
  if (qc < 128) qc = 128;
  const int64_t Bpad = (B + 127) / 128 * 128;
  return qc > Bpad ? Bpad : qc;
}
// The coarse level of the top-k threshold kernel (kernels_topk_bf16.hip: TK_GRP; round 5): one key per 16 visits of a thread, i.e.
// 1/16 of the block maxima of a query chunk -- bytes of the scratch for G blocks x qs queries, for either strip width (4 x QV
// queries, 512 / QV sub-ranges); 0: the block axis is too short for the route to pay.
constexpr int64_t TK_COARSE_MIN_BLOCKS = 16384;
inline size_t topk_coarse_bytes(int64_t G, int64_t qs) {
  if (G < TK_COARSE_MIN_BLOCKS) return 0;
  size_t need = 0;
  for (int QV : {4, 8}) {
    const int64_t SUB = 512 / QV, NG = ((G + SUB - 1) / SUB + 15) / 16, strips = (qs + 4 * QV - 1) / (4 * QV);
    const size_t b = (size_t)strips * NG * 512 * 16;
    need = b > need ? b : need;
  }
  return need;
}
// size of the block-grouped slot list: every block with candidates is padded to a multiple of 32
// counters per block for the grouping of candidate slots: few blocks = many slots per block = contended atomics
inline int topk_nseg(int64_t n_eblk) { return n_eblk < 4096 ? 8 : 1; }
// Large tables: the x3 count kernel writes one block maximum per 64 entities instead of 32 (kernels_score3_bf16.hip: GM = 2),
// the threshold kernel selects 64-entity candidate blocks and k_topk_expand64 turns each into its two 32-entity halves for
// everything downstream -- half the bytes of the three threshold sweeps against twice the re-scoring.  Measured crossover
// between WN18RR's 41 K entities (equal) and a 1.25 M-row shard (-0.3 ms of 7.1): from 65,536 local rows on.
inline int topk_expand(const coper_handle* h) {
  const char* force = getenv("COPER_TOPK_EXPAND");             // A/B and tests: "1" / "2" (read per call; set it before the handle's first top-k call and keep it)
  if (h->cfg.score_mode == COPER_SCORE_F32) return 1;
#if defined(COPER_SC3_MB) && COPER_SC3_MB != 4
  return 1;                                                    // (A/B builds with another block count per wave: no 64-entity form generated)
#endif
  if (h->dm.KS16 != 13 && h->dm.KS16 != 16) return 1;          // (the 64-entity form is instantiated for d = 200 and 256 only: build time)
  if (force && (force[0] == '1' || force[0] == '2')) return force[0] - '0';
  return h->dm.n_local >= 65536 ? 2 : 1;
}
inline int64_t topk_gm_rows(const coper_handle* h) { return h->dm.n_eblk / topk_expand(h); }     // rows of gmax (n_eblk is a multiple of 16)
inline size_t topk_sorted_cap(int64_t n_eblk, int64_t T) { return (size_t)((T + 31 * (n_eblk < T ? n_eblk : T) + 31) / 32 * 32); }
void score_count_begin_bf16x3(coper_handle* h, int64_t B, int32_t* ng, int32_t* ne, hipStream_t s);
// kernels_score3_bf16.hip: the count kernel (16x16x32, software-pipelined, one wave per SIMD) and the exact band
int score_count3_chunk_bf16x3(coper_handle* h, int64_t q0, int64_t Bc, const float* hvec, const float* tgt_x, const int64_t* e2,
                              const int64_t* indptr, const int64_t* idx, int32_t* ng, int32_t* ne, float* gmax, int64_t gm_stride,
                              hipStream_t s);
size_t score_count3_mask_bytes(const coper_handle* h, int64_t Bc);
int launch_band_consts(coper_handle* h, const float* ent, const float* bias, hipStream_t s);
int launch_band_setup(coper_handle* h, const float* hvec, const float* tgt, int64_t B, hipStream_t s);
int launch_exact_targets(coper_handle* h, const float* hvec, const int64_t* e2, int64_t B, float* out, hipStream_t s);
int launch_exact_rows(coper_handle* h, const float* hvec, const float* rows, const float* bias, int64_t B, float* out, hipStream_t s);
float band_kappa(const coper_handle* h);
// kernels_tail_bf16.hip: finalize + targets + filter correction of a ranking pass in one launch
bool tail_fused_supported(const coper_handle* h);
int launch_finalize_targets_filter_bf16x3(coper_handle* h, int64_t B, int ksplit, float* h_out, const int64_t* e2, const int64_t* indptr,
                                          const int64_t* idx, int64_t nnz, float* tgt, int32_t* ranks, hipStream_t s);
int launch_finalize_h_publish(coper_handle* h, int64_t B, int ksplit, float* h_out, hipStream_t s);
int launch_filter_excess_bf16x3(coper_handle* h, const float* hvec, const int64_t* e2, const int64_t* indptr, const int64_t* idx,
                                int64_t nnz, int64_t B, int32_t* ranks, bool defer, hipStream_t s);
void score_count_begin_f32(coper_handle* h, const float* hvec, int64_t B, int32_t* ng, int32_t* ne, hipStream_t s);
int score_count_chunk_f32(coper_handle* h, int64_t q0, int64_t Bc, const float* tgt, int32_t* ng, int32_t* ne, float* gmax,
                          int64_t gm_stride, hipStream_t s);
int launch_topk_score_blocks_f32(coper_handle* h, const float* hvec, int64_t T, const int64_t* e2, const int64_t* indptr,
                                 const int64_t* idx, hipStream_t s);
int launch_topk_pruned_f32(coper_handle* h, const float* hvec, const float* tgt, const int64_t* e2, const int64_t* indptr,
                           const int64_t* idx, int64_t nnz, int64_t B, int k, int32_t* ng, int32_t* ne, float* topk_val,
                           int64_t* topk_idx, hipStream_t s);
int launch_topk_pruned_bf16x3(coper_handle* h, const float* hvec, const float* tgt_x, const int64_t* e2, const int64_t* indptr,
                              const int64_t* idx, int64_t nnz, int64_t B, int k, int32_t* ng, int32_t* ne, float* topk_val,
                              int64_t* topk_idx, hipStream_t s);
// kernels_encode_bf16.hip
bool conv_bf16_supported(const Dims& dm);
int launch_wfrag_to_bf16(coper_handle* h, const float* Wf, int64_t Rw, void* hi, void* lo, hipStream_t s);
int compute_x_exp(coper_handle* h, unsigned* scratch, hipStream_t s);
// k-steps (of 32) between two feature blocks of the 16-bit dense-weight image Wf16_{hi,lo}: [rel * nfb + fb][stride][64] x 16 B.
// Round 5: ODD (F / 32 | 1: 145 for 144 at FB15k-237 shapes, +0.7 % memory).  The 26 streams of a workgroup and the 237 workgroups
// of a pass are all at the same k-step at the same time; with feature blocks 144 KiB and relations 1,872 KiB apart they all sat at
// the same offset inside every 4-KiB page of the image, i.e. on the same few memory channels.  Measured on the pass (same box,
// three alternating runs): fused encoder 0.1729 -> 0.1687 ms, pass 0.5115 -> 0.5064 ms.
__host__ __device__ inline int64_t w16_ks_stride_n(int64_t ks32n) {
  return ks32n | 1;
}
inline int64_t w16_ks_stride(const Dims& dm) { return w16_ks_stride_n(dm.F_pad / 32); }
int launch_conv_bf16(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, int64_t B,
                     bool skip_big, hipStream_t s);
int launch_dense_bf16(coper_handle* h, int64_t B, int nslices, bool small_only, hipStream_t s);
bool dense_fused_supported(const coper_handle* h, int nslices);
int launch_dense_fused_bf16(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, int64_t B,
                            int nslices, float* h_fin, hipStream_t s);
bool dense_fused_finalizes(const coper_handle* h, int nslices, const float* h_out);
int fused_fin_update(coper_handle* h, hipStream_t s);
// grouping sets (coper_group_next; coper_abi.hip)
void group_snapshot_home(coper_handle* h);          // after the home arrays were (re)allocated
void group_use_set(coper_handle* h, int i);         // point the fields at set i
void group_sets_free(coper_handle* h);              // sets 1, 2 (back to home first)
int group_sets_ensure(coper_handle* h, hipStream_t s);
int launch_dense_finalize(coper_handle* h, const int64_t* rel, int64_t B, int ksplit, float* h_out, hipStream_t s);
int launch_dense_finalize_pack(coper_handle* h, int64_t B, int ksplit, float* h_out, int32_t* cnt, int32_t cnt_base,
                               int32_t* cnt_eq, hipStream_t s);
int launch_pair_targets_packed_bf16x3(coper_handle* h, const int64_t* e2, int64_t B, float* tgt, hipStream_t s);
int score_all_dispatch(coper_handle* h, const float* hvec, int64_t B, float* logits, int64_t ld, hipStream_t s);
int launch_bias_pad(coper_handle* h, const float* bias, hipStream_t s);
int launch_topk(coper_handle* h, const float* hvec, const int64_t* e2, const int64_t* indptr, const int64_t* idx,
                int64_t B, int k, float* topk_val, int64_t* topk_idx, float* logits_ws, int64_t chunk_rows,
                hipStream_t s);

// COPER_DBG_SYNC=1 in the environment: synchronise after the launches that carry this hook and report the first failing one
// (localises a faulting kernel; read once per process)
int dbg_sync(coper_handle* h, hipStream_t s, const char* what);
#define COPER_DBG_SYNC(h, s, what) do { int _d = coper::dbg_sync((h), (s), (what)); if (_d) return _d; } while (0)

// profiling helpers (hipEvents on the launch stream)
struct ScopedKernelTimer {
  coper_handle* h;
  const char* name;
  hipStream_t s;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ScopedKernelTimer(coper_handle* h_, const char* n, hipStream_t s_);
  ~ScopedKernelTimer();
};

}  // namespace coper
