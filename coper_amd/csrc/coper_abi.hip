// libcoper_hip.so -- C ABI (include/coper_hip.h): handle, configuration validation, parameter
// registry, prepare, workspace, and dispatch to the kernels.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "bf16x3_chain.h"
#include "coper_internal.h"
#include <vector>
#include <mutex>
#include <unordered_map>

static thread_local std::string g_create_error;

namespace coper {

int fail(coper_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}

int hip_fail(coper_handle* h, hipError_t e, const char* what) {
  std::string msg = std::string("HIP error '") + hipGetErrorString(e) + "' in " + what;
  (void)hipGetLastError();
  return fail(h, COPER_EHIP, msg);
}

int dbg_sync(coper_handle* h, hipStream_t s, const char* what) {
  static const bool on = getenv("COPER_DBG_SYNC") != nullptr;
  if (!on) return COPER_OK;
  fprintf(stderr, "[coper] sync after %s ... ", what);
  fflush(stderr);
  const hipError_t e = hipStreamSynchronize(s);
  fprintf(stderr, "%s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
  if (e != hipSuccess) return hip_fail(h, e, what);
  return COPER_OK;
}

// Event pairs come from a pool (creating events costs tens of microseconds of host time per launch: inside bench.py's
// timed region that alone was 10 % of a pass) and carry no system-scope release (device-side timestamps only).
static hipEvent_t timer_event(coper_handle* h) {
  if (!h->event_pool.empty()) {
    hipEvent_t e = h->event_pool.back();
    h->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventReleaseToDevice) != hipSuccess) {
    (void)hipGetLastError();
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  }
  return e;
}

static std::mutex g_ledger_mu;
static std::unordered_map<void*, size_t> g_ledger;
static int64_t g_ledger_bytes = 0;

// COPER_DBG_REDZONE=<bytes> (round 6): every allocation of the library is followed by that many bytes of 0xA5; when it is freed the
// bytes are read back and a changed one ends the process with the allocation's size -- a kernel that writes past the end of a
// workspace fails the first test that frees it instead of whichever test owns the neighbouring memory (found the way this was built:
// a workspace of the training step sized for one of its two uses, overrun for 12 rounds of tests that happened not to notice).
static size_t redzone_bytes() {
  static const char* rz = getenv("COPER_DBG_REDZONE");
  static const size_t n = rz ? (size_t)strtoull(rz, nullptr, 0) : 0;
  return n;
}

hipError_t tracked_malloc_impl(void** p, size_t bytes) {
  const size_t rz = redzone_bytes();
  const hipError_t e = hipMalloc(p, bytes + rz);
  if (e == hipSuccess && *p && rz) (void)hipMemset((char*)*p + bytes, 0xA5, rz);
  // COPER_DBG_POISON=<byte>: every allocation of the library starts filled with that byte (0xFF: NaN patterns) -- a kernel that
  // reads workspace nobody wrote shows up as a wrong result in the first test that runs, not in whichever test inherits a
  // predecessor's memory (tools/README.md)
  // COPER_DBG_POISON_ONLY=<k>,<byte>: allocation number k of the process gets <byte> instead (and says so): which buffer is it
  static const char* poison = getenv("COPER_DBG_POISON");
  static const char* only = getenv("COPER_DBG_POISON_ONLY");
  static long n_alloc = 0;
  if (e == hipSuccess && *p && poison) {
    int byte = (int)strtol(poison, nullptr, 0);
    if (only) {
      char* end = nullptr;
      const long k = strtol(only, &end, 0);
      if (k == n_alloc && end && *end == ',') {
        byte = (int)strtol(end + 1, nullptr, 0);
        fprintf(stderr, "[coper] allocation %ld (%zu bytes) poisoned with 0x%02x\n", k, bytes, byte & 255);
      }
    }
    ++n_alloc;
    (void)hipMemset(*p, byte, bytes);
  }
  if (e == hipSuccess && *p) {
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    g_ledger[*p] = bytes;
    g_ledger_bytes += (int64_t)bytes;
  }
  return e;
}

hipError_t tracked_free(void* p) {
  if (!p) return hipSuccess;
  size_t bytes = 0;
  bool known = false;
  {
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    auto it = g_ledger.find(p);
    if (it != g_ledger.end()) {
      bytes = it->second;
      known = true;
      g_ledger_bytes -= (int64_t)it->second;
      g_ledger.erase(it);
    }
  }
  const size_t rz = redzone_bytes();
  if (known && rz) {
    std::vector<unsigned char> tail(rz);
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(tail.data(), (char*)p + bytes, rz, hipMemcpyDeviceToHost) == hipSuccess) {
      for (size_t i = 0; i < rz; ++i)
        if (tail[i] != 0xA5) {
          fprintf(stderr, "[coper] COPER_DBG_REDZONE: byte %zu behind an allocation of %zu bytes was overwritten (0x%02x)\n", i, bytes, tail[i]);
          abort();
        }
    } else {
      (void)hipGetLastError();
    }
  }
  return hipFree(p);
}

ScopedKernelTimer::ScopedKernelTimer(coper_handle* h_, const char* n, hipStream_t s_) : h(h_), name(n), s(s_) {
  if (!h->profile) return;
  {   // a pass being captured into a hipGraph is not timed: events recorded into a capture cannot be read back (found by a test that
      // left the profile on while capturing: hipEventElapsedTime -> invalid resource handle at the next coper_profile_read)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cap != hipStreamCaptureStatusNone) return;
  }
  e0 = timer_event(h);
  e1 = timer_event(h);
  if (!e0 || !e1) { e0 = e1 = nullptr; return; }
  (void)hipEventRecord(e0, s);
}

ScopedKernelTimer::~ScopedKernelTimer() {
  if (!e0) return;
  (void)hipEventRecord(e1, s);
  h->timers[name].pending.emplace_back(e0, e1);
}

static void add_spec(coper_handle* h, const std::string& name, std::vector<int64_t> shape) {
  h->specs.push_back({name, shape});
  h->params[name] = Param();
}

static void add_bn_specs(coper_handle* h, const std::string& prefix, int64_t n) {
  add_spec(h, prefix + "/gamma", {n});
  add_spec(h, prefix + "/beta", {n});
  add_spec(h, prefix + "/moving_mean", {n});
  add_spec(h, prefix + "/moving_variance", {n});
}

// ContextualParameterGenerator.__init__ (models.py:44-54): projections [in, n] over context_size[1:] + [num_elements]
static void add_generator_specs(coper_handle* h, const std::string& name, int n_hidden, const int32_t* hidden,
                                int64_t num_elements) {
  int64_t in = h->dm.r;
  for (int i = 0; i <= n_hidden; ++i) {
    int64_t n = i < n_hidden ? hidden[i] : num_elements;
    char buf[256];
    snprintf(buf, sizeof buf, "%s/CPG/Projection%d", name.c_str(), i);
    add_spec(h, buf, {in, n});
    if (i < n_hidden && h->dm.ctx_bn) add_bn_specs(h, std::string(buf) + "/BatchNorm", n);
    in = n;
  }
}

static int64_t prod(const std::vector<int64_t>& v) {
  int64_t p = 1;
  for (auto x : v) p *= x;
  return p;
}

template <typename T>
static int dev_alloc(coper_handle* h, T** p, size_t n) {
  if (*p) { (void)tracked_free(*p); *p = nullptr; }
  if (n == 0) n = 1;
  hipError_t e = tracked_malloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) {
    *p = nullptr;
    (void)hipGetLastError();
    char buf[128];
    snprintf(buf, sizeof buf, "hipMalloc of %zu bytes failed", n * sizeof(T));
    return fail(h, COPER_ENOMEM, buf);
  }
  return COPER_OK;
}

template <typename T>
static void dev_free(T** p) {
  if (*p) { (void)tracked_free(*p); *p = nullptr; }
}

// ---- grouping sets (coper_group_next) ----
void group_snapshot_home(coper_handle* h) {
  coper_handle::GroupSet& g = h->gset[0];
  g.rel_count = h->rel_count; g.rel_offset = h->rel_offset; g.perm = h->perm; g.inv_perm = h->inv_perm; g.sorted_row = h->sorted_row;
  g.sorted_rid = h->sorted_rid; g.tiles = h->tiles; g.n_tiles = h->n_tiles; g.x3m = h->x3m; g.fused_fin_dev = h->fused_fin_dev;
  g.fused_fin_perm = h->fused_fin_perm;
}

void group_use_set(coper_handle* h, int i) {
  if (h->gcur == i) return;
  if (h->gcur == 0) group_snapshot_home(h);
  const coper_handle::GroupSet& g = h->gset[i];
  h->rel_count = g.rel_count; h->rel_offset = g.rel_offset; h->perm = g.perm; h->inv_perm = g.inv_perm; h->sorted_row = g.sorted_row;
  h->sorted_rid = g.sorted_rid; h->tiles = g.tiles; h->n_tiles = g.n_tiles; h->x3m = g.x3m; h->fused_fin_dev = g.fused_fin_dev;
  h->fused_fin_perm = g.fused_fin_perm;
  h->gcur = i;
}

void group_sets_free(coper_handle* h) {
  group_use_set(h, 0);
  for (int i = 1; i < 3; ++i) {
    coper_handle::GroupSet& g = h->gset[i];
    if (g.slab) (void)tracked_free(g.slab);
    if (g.fused_fin_dev) (void)tracked_free(g.fused_fin_dev);
    g = coper_handle::GroupSet();
  }
  h->pipe.invalidate_grouping();
  h->pass_chk = nullptr;
}

// sets 1 and 2 at the capacity of the workspace (which exists: the caller is a pass being enqueued), with their finalize constants
int group_sets_ensure(coper_handle* h, hipStream_t s) {
  if (h->gset[1].slab && h->gset[2].slab) return COPER_OK;
  const Dims& dm = h->dm;
  if (!h->perm || !h->x3m || h->ws_queries <= 0) return fail(h, COPER_ESTATE, "grouping sets: no workspace yet");
  const int cur = h->gcur;
  group_use_set(h, 0);
  const size_t cap = (size_t)h->ws_queries, r2 = (size_t)dm.R + 2, nt = 4 * (cap / 32 + (size_t)dm.R + 4);
  auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };      // (16-byte pieces)
  const size_t total = 2 * up4(r2) + 4 * up4(cap) + up4(nt) + 4 + (size_t)X3M_SLOTS + 2 * GROUP_CHK_WORDS;
  for (int i = 1; i < 3; ++i) {
    coper_handle::GroupSet& g = h->gset[i];
    if (g.slab) continue;
    if (tracked_malloc(&g.slab, total * sizeof(int32_t)) != hipSuccess) { (void)hipGetLastError(); return fail(h, COPER_ENOMEM, "hipMalloc failed (grouping set)"); }
    COPER_HIP_TRY(h, hipMemsetAsync(g.slab, 0, total * sizeof(int32_t), s));
    int32_t* p = g.slab;
    g.rel_count = p; p += up4(r2);
    g.rel_offset = p; p += up4(r2);
    g.perm = p; p += up4(cap);
    g.inv_perm = p; p += up4(cap);
    g.sorted_row = p; p += up4(cap);
    g.sorted_rid = p; p += up4(cap);
    g.tiles = p; p += up4(nt);
    g.n_tiles = p; p += 4;
    g.x3m = (float*)p; p += X3M_SLOTS;       // (a multiple of four words: the check words below are 8-byte aligned)
    g.chk = (int64_t*)p;
  }
  int rc = fused_fin_update(h, s);
  group_use_set(h, cur);
  return rc;
}

static int ensure_workspace(coper_handle* h, int64_t B, int64_t nnz, hipStream_t s) {
  const Dims& dm = h->dm;
  if (nnz > h->ws_nnz) h->ws_nnz = nnz;
  if (h->cfg.score_mode != COPER_SCORE_F32 && nnz > h->row_of_cap) {
    COPER_HIP_TRY(h, hipStreamSynchronize(s));
    int rc0 = dev_alloc(h, &h->row_of_ws, (size_t)nnz);
    if (rc0) return rc0;
    h->row_of_cap = nnz;
  }
  if (B <= h->ws_queries && h->perm) return COPER_OK;
  COPER_HIP_TRY(h, hipStreamSynchronize(s));
  if (h->group_done) {          // (the device counter of coper_stale_passes lives in the slab replaced below)
    int32_t v = 0;
    COPER_HIP_TRY(h, hipMemcpy(&v, h->group_done + 2, sizeof v, hipMemcpyDeviceToHost));
    h->stale_passes_host += v;
  }
  group_sets_free(h);           // (sized by the workspace; back on the home set before its arrays move)
  int64_t cap = B < 64 ? 64 : B;
  int rc;
  const int KSPLIT_MAX = 8;
  // published counts | accumulation buffer (see launch_group_by_relation) | cursors | ticket; zeroed once here, kept zero by the kernels
  dev_free(&h->rel_count_buf[0]);
  if ((rc = dev_alloc(h, &h->rel_count_buf[0], 3 * (dm.R + 2) + 4))) return rc;
  COPER_HIP_TRY(h, hipMemsetAsync(h->rel_count_buf[0], 0, sizeof(int32_t) * (3 * (dm.R + 2) + 4), s));
  h->rel_count_buf[1] = h->rel_count_buf[0] + (dm.R + 2);
  h->rel_cursor = h->rel_count_buf[0] + 2 * (dm.R + 2);
  h->group_done = h->rel_count_buf[0] + 3 * (dm.R + 2);
  h->rel_count = h->rel_count_buf[0];
  if ((rc = dev_alloc(h, &h->rel_offset, dm.R + 2))) return rc;
  if ((rc = dev_alloc(h, &h->perm, cap))) return rc;
  if ((rc = dev_alloc(h, &h->inv_perm, cap))) return rc;
  if ((rc = dev_alloc(h, &h->sorted_row, cap))) return rc;
  if ((rc = dev_alloc(h, &h->sorted_rid, cap))) return rc;
  if ((rc = dev_alloc(h, &h->tiles, 4 * (cap / 32 + dm.R + 4)))) return rc;
  if ((rc = dev_alloc(h, &h->n_tiles, 4))) return rc;
  if ((rc = dev_alloc(h, &h->blk_off, dm.R + 2))) return rc;
  if ((rc = dev_alloc(h, &h->x_sorted, (size_t)cap * dm.F_pad))) return rc;
  if ((rc = dev_alloc(h, &h->z_part, (size_t)KSPLIT_MAX * cap * dm.d_pad16))) return rc;
  if ((rc = dev_alloc(h, &h->tgt_ws, 2 * cap))) return rc;    // [mode logit | exact-chain logit] (coper_target_scores)
  if ((rc = dev_alloc(h, &h->cnt_ws, 2 * cap))) return rc;
  if (h->cfg.score_mode == COPER_SCORE_F32) {
    if ((rc = dev_alloc(h, &h->hfrag_ws, (size_t)((cap + 127) / 128) * 128 * dm.d_pad8))) return rc;
  } else {
    size_t plane = (size_t)((cap + 127) / 128) * 4 * dm.KS16 * 64 * 16;
    const size_t f3 = (size_t)((cap + 127) / 128) * 8 * f3_steps(dm.KS16) * 2 * 64 * 16;   // the count kernel's query image
    dev_free((char**)&h->hfrag16_hi); dev_free((char**)&h->hfrag16_lo);
    dev_free((char**)&h->hrm16_hi); dev_free((char**)&h->hrm16_lo); dev_free((char**)&h->hf3_ws);
    if (tracked_malloc(&h->hfrag16_hi, plane) != hipSuccess || tracked_malloc(&h->hfrag16_lo, plane) != hipSuccess ||
        tracked_malloc(&h->hrm16_hi, plane) != hipSuccess || tracked_malloc(&h->hrm16_lo, plane) != hipSuccess ||
        tracked_malloc(&h->hf3_ws, f3) != hipSuccess)
      return fail(h, COPER_ENOMEM, "hipMalloc of the bf16 query planes failed");
    COPER_HIP_TRY(h, hipMemsetAsync(h->hf3_ws, 0, f3, s));     // the zero halves of its tail registers are never written again
    if ((rc = dev_alloc(h, &h->tband_ws, 2 * (size_t)((cap + 127) / 128 * 128))) || (rc = dev_alloc(h, &h->tgtx_ws, cap))) return rc;
    const size_t n_heavy = (size_t)(cap / 32 + 8);
    if ((rc = dev_alloc(h, &h->heavy_ws, n_heavy))) return rc;
    COPER_HIP_TRY(h, hipMemsetAsync(h->heavy_ws, 0, sizeof(int32_t) * n_heavy, s));   // the excess kernel leaves it zero again
  }
  h->ws_queries = cap;
  h->ws_ksplit = KSPLIT_MAX;
  group_snapshot_home(h);
  return fused_fin_update(h, s);        // (perm moved)
}

// what only the ranking entry points need, on top of ensure_workspace: the count kernel's band mask (1 bit per logit of
// a count launch), the exact targets and the internal fp32 h of coper_encode_rank -- not allocated for encode / score_all
// callers (the mask of a 4,096-query launch against 10 M entities is 5 GB)
static int ensure_rank_workspace(coper_handle* h, int64_t B, int64_t nnz, bool need_h, hipStream_t s) {
  int rc;
  if ((rc = ensure_workspace(h, B, nnz, s))) return rc;
  if (need_h && (!h->h_ws || h->h_ws_rows < B)) {
    COPER_HIP_TRY(h, hipStreamSynchronize(s));
    const int64_t rows = B > h->ws_queries ? B : h->ws_queries;
    if ((rc = dev_alloc(h, &h->h_ws, (size_t)rows * h->dm.d))) return rc;
    h->h_ws_rows = rows;
  }
  if (h->cfg.score_mode == COPER_SCORE_F32) return COPER_OK;
  // the longest count launch any path issues: the mask has the size of the block maxima of the pruned top-k (one bit per logit
  // against one float per 32), so both are cut into the same chunks of queries
  const int64_t qc = topk_chunk_queries(h->dm.n_eblk, B, h->gmax_max_floats);
  const size_t need = score_count3_mask_bytes(h, qc);
  if (need > h->mask_cap) {
    COPER_HIP_TRY(h, hipStreamSynchronize(s));
    dev_free((char**)&h->mask_ws);
    h->mask_cap = 0;
    if (tracked_malloc(&h->mask_ws, need) != hipSuccess) { (void)hipGetLastError(); return fail(h, COPER_ENOMEM, "hipMalloc of the band mask failed"); }
    h->mask_cap = need;
  }
  return COPER_OK;
}

int score_all_dispatch(coper_handle* h, const float* hvec, int64_t B, float* logits, int64_t ld, hipStream_t s) {
  if (h->cfg.score_mode != COPER_SCORE_F32) return launch_score_all_bf16x3(h, hvec, B, logits, ld, s);
  return launch_score_all(h, hvec, B, logits, ld, s);
}

}  // namespace coper

using namespace coper;

extern "C" {

COPER_API int coper_abi_version(void) { return COPER_ABI_VERSION; }

COPER_API const char* coper_last_error(const coper_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

COPER_API int coper_create(const coper_config* cfg, coper_handle** out) {
  if (!cfg || !out) return fail(nullptr, COPER_EINVAL, "coper_create: null argument");
  *out = nullptr;
  if (cfg->abi_version != COPER_ABI_VERSION) return fail(nullptr, COPER_EINVAL, "coper_create: abi_version mismatch");
  coper_handle* h = new coper_handle();
  h->cfg = *cfg;
  Dims& dm = h->dm;
  auto bad = [&](const char* m) {
    int rc = fail(nullptr, COPER_EINVAL, std::string("coper_create: ") + m);
    delete h;
    return rc;
  };
  dm.E = cfg->num_ent; dm.R = cfg->num_rel; dm.d = cfg->ent_emb_size; dm.r = cfg->rel_emb_size;
  dm.emb_h = cfg->emb_h; dm.emb_w = cfg->emb_w;
  dm.fh = cfg->conv_filter_height; dm.fw = cfg->conv_filter_width; dm.C = cfg->conv_num_channels;
  dm.concat_rel = cfg->concat_rel != 0; dm.lookup = cfg->do_parameter_lookup != 0;
  dm.gen_conv = cfg->n_ctx_conv >= 0; dm.gen_fc = cfg->n_ctx_out >= 0;
  dm.ctx_bn = cfg->context_rel_use_batch_norm != 0;
  if (dm.E <= 0 || dm.R <= 0 || dm.d <= 0 || dm.r <= 0) return bad("num_ent, num_rel, ent_emb_size, rel_emb_size must be positive");
  if (dm.R > 0x7fffffff || dm.E > 0x7fffffffffLL) return bad("num_rel / num_ent too large");
  if (dm.fh <= 0 || dm.fw <= 0 || dm.C <= 0) return bad("conv filter sizes must be positive");
  if (dm.emb_h <= 0 || dm.emb_w <= 0 || dm.emb_h * dm.emb_w != dm.d) return bad("emb_h * emb_w must equal ent_emb_size (models.py:355)");
  if (cfg->n_ctx_conv > COPER_MAX_CTX || cfg->n_ctx_out > COPER_MAX_CTX) return bad("too many generator hidden layers");
  if (cfg->shard_lo < 0 || cfg->shard_hi > dm.E || cfg->shard_lo >= cfg->shard_hi) return bad("entity shard [lo,hi) out of range");
  if (cfg->score_mode != COPER_SCORE_F32 && cfg->score_mode != COPER_SCORE_BF16X3)
    return bad("score_mode: COPER_SCORE_F32 or COPER_SCORE_BF16X3");
  if (cfg->role != COPER_ROLE_BOTH && cfg->role != COPER_ROLE_ENCODE && cfg->role != COPER_ROLE_SCORE)
    return bad("role: COPER_ROLE_BOTH, COPER_ROLE_ENCODE or COPER_ROLE_SCORE");
  if (cfg->rel_mod_world < 0 || (cfg->rel_mod_world > 1 && (cfg->rel_mod_rank < 0 || cfg->rel_mod_rank >= cfg->rel_mod_world)))
    return bad("rel_mod_world >= 0 and 0 <= rel_mod_rank < rel_mod_world");
  // models.py:360: e1 stacked on the reshaped relation only for plain ConvE
  dm.stacked = !dm.gen_conv && !dm.gen_fc && !dm.lookup;
  dm.in_h = dm.emb_h; dm.in_w = dm.emb_w;
  if (dm.stacked) {
    if (dm.r % dm.emb_h != 0 || dm.r / dm.emb_h != dm.emb_w)
      return bad("plain ConvE stacks e1 on rel (models.py:361-362): needs rel_emb_size == ent_emb_size");
    dm.in_h = 2 * dm.emb_h;
  } else if (!dm.gen_conv && !dm.gen_fc) {
    return bad("do_parameter_lookup with both contexts None is ill-formed in the reference (models.py:263-264 vs :360)");
  }
  if (dm.lookup && dm.concat_rel) return bad("g_lookup passes relation ids as rel_emb: cannot concat_rel (models.py:180,406)");
  dm.Ho = dm.in_h - dm.fh + 1; dm.Wo = dm.in_w - dm.fw + 1;
  if (dm.Ho <= 0 || dm.Wo <= 0) return bad("conv filter larger than the image");
  dm.F_conv = (int64_t)dm.Ho * dm.Wo * dm.C;
  dm.F = dm.F_conv + (dm.concat_rel ? dm.r : 0);
  dm.F_pad = (dm.F + 31) / 32 * 32;
  dm.d_pad16 = (dm.d + 15) / 16 * 16; dm.nfb = dm.d_pad16 / 16;
  dm.d_pad8 = (dm.d + 7) / 8 * 8; dm.KS = dm.d_pad8 / 8;
  dm.KS16 = (dm.d + 15) / 16;
  dm.n_local = cfg->shard_hi - cfg->shard_lo;
  dm.n_eblk = ((dm.n_local + 31) / 32 + EBLK_ALIGN - 1) / EBLK_ALIGN * EBLK_ALIGN;
  if (dm.KS * 4 * 64 * 16 > 160 * 1024) return bad("ent_emb_size too large for the LDS query tile (d <= 320)");
  if ((int64_t)dm.in_h * dm.in_w + (int64_t)dm.fh * dm.fw * dm.C + 3 * dm.C > 40000) return bad("conv stage too large for LDS");

  // parameter specs (models.py:203-336)
  add_spec(h, "ent_emb", {dm.n_local, dm.d});
  add_spec(h, "pred_bias", {dm.n_local});
  if (!dm.lookup) add_spec(h, "rel_emb", {dm.R, dm.r});
  int64_t nconv = (int64_t)dm.fh * dm.fw * dm.C;
  if (dm.gen_conv) {
    if (dm.lookup) {
      add_spec(h, "conv1_weights", {dm.R, nconv});
      add_spec(h, "conv1_bias", {dm.R, dm.C});
    } else {
      add_generator_specs(h, "conv1_weights", cfg->n_ctx_conv, cfg->ctx_conv, nconv);
      add_generator_specs(h, "conv1_bias", cfg->n_ctx_conv, cfg->ctx_conv, dm.C);
    }
  } else {
    add_spec(h, "conv1_weights", {dm.fh, dm.fw, 1, dm.C});
    add_spec(h, "conv1_bias", {dm.C});
  }
  if (dm.gen_fc) {
    if (dm.lookup) {
      add_spec(h, "fc_weights", {dm.R, dm.F * dm.d});
      add_spec(h, "fc_bias", {dm.R, dm.d});
    } else {
      add_generator_specs(h, "fc_weights", cfg->n_ctx_out, cfg->ctx_out, dm.F * dm.d);
      add_generator_specs(h, "fc_bias", cfg->n_ctx_out, cfg->ctx_out, dm.d);
    }
  } else {
    add_spec(h, "fc_weights", {dm.F, dm.d});
    add_spec(h, "fc_bias", {dm.d});
  }
  add_bn_specs(h, "Conv1BN", dm.C);
  add_bn_specs(h, "FCBN", dm.d);
  *out = h;
  return COPER_OK;
}

COPER_API void coper_destroy(coper_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  (void)hipDeviceSynchronize();
  train_destroy(h);
  group_sets_free(h);
  dev_free(&h->conv_scale); dev_free(&h->conv_shift); dev_free(&h->fc_scale); dev_free(&h->fc_shift);
  dev_free(&h->conv_w_rel); dev_free(&h->conv_b_rel); dev_free(&h->fc_b_rel); dev_free(&h->Wf);
  dev_free(&h->Ef); dev_free(&h->bias_pad); dev_free(&h->ctx_tmp[0]); dev_free(&h->ctx_tmp[1]);
  dev_free(&h->rel_count_buf[0]); h->rel_count = nullptr; h->rel_count_buf[1] = nullptr; h->group_done = nullptr; dev_free(&h->rel_offset); h->rel_cursor = nullptr; dev_free(&h->perm); dev_free(&h->inv_perm); dev_free(&h->sorted_row); dev_free(&h->sorted_rid);
  dev_free(&h->tiles); dev_free(&h->n_tiles); dev_free(&h->blk_off); dev_free(&h->x_sorted); dev_free(&h->z_part);
  dev_free(&h->tgt_ws); dev_free(&h->h_ws); dev_free(&h->cnt_ws); dev_free(&h->hfrag_ws); dev_free(&h->logits_ws); dev_free(&h->row_of_ws);
  dev_free(&h->tk_coarse_ws); dev_free(&h->gmax_ws); dev_free(&h->cand_blk_ws); dev_free(&h->cand_val_ws); dev_free(&h->cand_q_ws); dev_free(&h->cand_tau_ws); dev_free(&h->cand_sorted_ws); dev_free(&h->blk_cnt_ws); dev_free(&h->blk_off_ws);
  dev_free((char**)&h->Wf16_hi); dev_free((char**)&h->Wf16_lo);
  dev_free((char**)&h->Ef16_hi); dev_free((char**)&h->Ef16_lo); dev_free((char**)&h->hfrag16_hi); dev_free((char**)&h->hfrag16_lo);
  dev_free((char**)&h->Erm16_hi); dev_free((char**)&h->Erm16_lo); dev_free((char**)&h->hrm16_hi); dev_free((char**)&h->hrm16_lo);
  dev_free((char**)&h->Ef3); dev_free((char**)&h->hf3_ws); dev_free((char**)&h->mask_ws); dev_free(&h->band_consts); dev_free(&h->tband_ws); dev_free(&h->x3s); dev_free(&h->x3m); dev_free(&h->w_exp); dev_free((char**)&h->fused_fin_dev);
  dev_free(&h->tgtx_ws); dev_free(&h->heavy_ws);
  for (auto& kv : h->timers)
    for (auto& p : kv.second.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  for (auto e : h->event_pool) (void)hipEventDestroy(e);
  delete h;
}

COPER_API int coper_get_dims(const coper_handle* h, int64_t* F, int32_t* Ho, int32_t* Wo, int64_t* n_local) {
  if (!h) return COPER_EINVAL;
  if (F) *F = h->dm.F;
  if (Ho) *Ho = h->dm.Ho;
  if (Wo) *Wo = h->dm.Wo;
  if (n_local) *n_local = h->dm.n_local;
  return COPER_OK;
}

COPER_API int coper_num_params(const coper_handle* h) { return h ? (int)h->specs.size() : 0; }

COPER_API int coper_param_spec(const coper_handle* h, int index, const char** leaf_name, int64_t* shape_out, int* ndim_out) {
  if (!h || index < 0 || index >= (int)h->specs.size()) return COPER_EINVAL;
  const ParamSpec& sp = h->specs[index];
  if (leaf_name) *leaf_name = sp.name.c_str();
  if (ndim_out) *ndim_out = (int)sp.shape.size();
  if (shape_out)
    for (size_t i = 0; i < sp.shape.size() && i < 4; ++i) shape_out[i] = sp.shape[i];
  return COPER_OK;
}

COPER_API int coper_set_param(coper_handle* h, const char* leaf_name, const void* dev_ptr, const int64_t* shape, int ndim) {
  if (!h || !leaf_name || !dev_ptr || !shape || ndim <= 0) return fail(h, COPER_EINVAL, "coper_set_param: null argument");
  auto it = h->params.find(leaf_name);
  if (it == h->params.end())
    return fail(h, COPER_EINVAL, std::string("coper_set_param: '") + leaf_name + "' is not a parameter of this configuration");
  const ParamSpec* sp = nullptr;
  for (auto& s : h->specs)
    if (s.name == leaf_name) sp = &s;
  std::vector<int64_t> got(shape, shape + ndim);
  if (prod(got) != prod(sp->shape)) {
    char buf[256];
    snprintf(buf, sizeof buf, "coper_set_param: '%s' has %lld elements, configuration needs %lld", leaf_name,
             (long long)prod(got), (long long)prod(sp->shape));
    return fail(h, COPER_ESHAPE, buf);
  }
  it->second.ptr = (const float*)dev_ptr;
  it->second.shape = got;
  it->second.set = true;
  h->prepared = false;
  train_params_changed(h);
  return COPER_OK;
}

// Evaluate one generator for every relation id: ctx chain through the hidden layers (BN folded, ReLU),
// returns the final context and its width.  (models.py:56-70)
static int run_generator_hidden(coper_handle* h, const std::string& name, int n_hidden, const int32_t* hidden,
                                const float** ctx_out, int* K_out, hipStream_t s) {
  const Dims& dm = h->dm;
  const float* cur = h->params["rel_emb"].ptr;
  int K = dm.r;
  for (int i = 0; i < n_hidden; ++i) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s/CPG/Projection%d", name.c_str(), i);
    int n = hidden[i];
    float* dst = h->ctx_tmp[i & 1];
    const float* sc = nullptr;
    const float* sh = nullptr;
    float* scb = nullptr;
    if (dm.ctx_bn) {
      // fold this layer's BN into a scratch (scale | shift) placed after the activations
      scb = h->ctx_tmp[i & 1] + (size_t)dm.R * n;
      std::string bn = std::string(buf) + "/BatchNorm";
      int rc = launch_fold_bn(h, h->params[bn + "/gamma"].ptr, h->params[bn + "/beta"].ptr,
                              h->params[bn + "/moving_mean"].ptr, h->params[bn + "/moving_variance"].ptr, n,
                              h->cfg.bn_epsilon, scb, scb + n, s);
      if (rc) return rc;
      sc = scb; sh = scb + n;
    }
    int rc = launch_gen_small(h, cur, dm.R, K, h->params[buf].ptr, n, sc, sh, true, dst, s);
    if (rc) return rc;
    cur = dst;
    K = n;
  }
  *ctx_out = cur;
  *K_out = K;
  return COPER_OK;
}

COPER_API int coper_set_x3_ent_absmax(coper_handle* h, float absmax) {
  if (!h || !(absmax >= 0.f)) return fail(h, COPER_EINVAL, "coper_set_x3_ent_absmax: bad argument");
  h->cfg.x3_ent_absmax = absmax;
  h->prepared = false;     // the entity planes were built for another power of two
  return COPER_OK;
}

COPER_API int coper_prepare(coper_handle* h, void* stream) {
  if (!h) return COPER_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const Dims& dm = h->dm;
  const coper_config& cfg = h->cfg;
  for (auto& sp : h->specs)
    if (!h->params[sp.name].set) return fail(h, COPER_EMISSING, "coper_prepare: parameter '" + sp.name + "' was never set");
  COPER_HIP_TRY(h, hipSetDevice(cfg.device));
  int rc;
  group_use_set(h, 0);          // (x3m of the home set is re-allocated below; a grouping done ahead does not survive a prepare)
  h->pipe.invalidate_grouping();
  if ((rc = dev_alloc(h, &h->conv_scale, dm.C)) || (rc = dev_alloc(h, &h->conv_shift, dm.C)) ||
      (rc = dev_alloc(h, &h->fc_scale, dm.d)) || (rc = dev_alloc(h, &h->fc_shift, dm.d)))
    return rc;
  auto P = [&](const std::string& n) { return h->params[n].ptr; };
  if ((rc = launch_fold_bn(h, P("Conv1BN/gamma"), P("Conv1BN/beta"), P("Conv1BN/moving_mean"),
                           P("Conv1BN/moving_variance"), dm.C, cfg.bn_epsilon, h->conv_scale, h->conv_shift, s)))
    return rc;
  if ((rc = launch_fold_bn(h, P("FCBN/gamma"), P("FCBN/beta"), P("FCBN/moving_mean"), P("FCBN/moving_variance"),
                           dm.d, cfg.bn_epsilon, h->fc_scale, h->fc_shift, s)))
    return rc;

  if (cfg.score_mode != COPER_SCORE_F32) {
    // the shard's maxima (row norm, |pred_bias|: the exact band; |element|: the power of two of the entity planes, split16.h)
    if ((rc = dev_alloc(h, &h->band_consts, BAND_NCONST)) || (rc = dev_alloc(h, &h->x3s, 4)) ||
        (rc = dev_alloc(h, &h->x3m, (size_t)X3M_SLOTS)))
      return rc;
    COPER_HIP_TRY(h, hipMemsetAsync(h->x3s, 0, 4 * sizeof(int32_t), s));
    COPER_HIP_TRY(h, hipMemsetAsync(h->x3m, 0, X3M_SLOTS * sizeof(float), s));
    group_snapshot_home(h);
    if ((rc = launch_band_consts(h, P("ent_emb"), P("pred_bias"), s))) return rc;
    {
      unsigned cb[BAND_NCONST];
      COPER_HIP_TRY(h, hipMemcpyAsync(cb, h->band_consts, sizeof cb, hipMemcpyDeviceToHost, s));
      COPER_HIP_TRY(h, hipStreamSynchronize(s));
      float xmax;
      memcpy(&xmax, &cb[2], sizeof xmax);
      // shards of one table agree on the exponent through the hint (the mode's logits are then the same bits on every shard
      // layout); a hint below the shard's own maximum could overflow fp16: refused
      if (cfg.x3_ent_absmax > 0.f) {
        if (!(cfg.x3_ent_absmax >= xmax)) return fail(h, COPER_EINVAL, "coper_prepare: x3_ent_absmax is below the largest |ent_emb| element of the shard");
        xmax = cfg.x3_ent_absmax;
      }
      unsigned xb;
      memcpy(&xb, &xmax, sizeof xb);
      h->x3_ent_absmax = xmax;
      h->x3_ent_exp = x3_exp_for_bits(xb);
    }
  }

  const bool role_enc = cfg.role != COPER_ROLE_SCORE, role_score = cfg.role != COPER_ROLE_ENCODE;
  h->enc_bf16 = false;
  h->x_exp = 0;
  h->band_launches = 0;
  h->Rw = 0;
  if (role_enc) {      // ---- the encoder's derived state (a COPER_ROLE_SCORE handle has none: no generator evaluated, no W_r cached)
  // scratch for generator hidden activations: R * max_hidden (+ 2 * max_hidden for the folded BN)
  int max_hidden = 1;
  for (int i = 0; i < cfg.n_ctx_conv; ++i) max_hidden = cfg.ctx_conv[i] > max_hidden ? cfg.ctx_conv[i] : max_hidden;
  for (int i = 0; i < cfg.n_ctx_out; ++i) max_hidden = cfg.ctx_out[i] > max_hidden ? cfg.ctx_out[i] : max_hidden;
  size_t tmp_elems = (size_t)(dm.R + 2) * max_hidden;
  if ((rc = dev_alloc(h, &h->ctx_tmp[0], tmp_elems)) || (rc = dev_alloc(h, &h->ctx_tmp[1], tmp_elems))) return rc;

  int64_t nconv = (int64_t)dm.fh * dm.fw * dm.C;
  if (dm.gen_conv) {
    if ((rc = dev_alloc(h, &h->conv_w_rel, (size_t)dm.R * nconv)) || (rc = dev_alloc(h, &h->conv_b_rel, (size_t)dm.R * dm.C)))
      return rc;
    if (dm.lookup) {  // ParameterLookup.generate (models.py:90-94): the table row IS the parameter
      COPER_HIP_TRY(h, hipMemcpyAsync(h->conv_w_rel, P("conv1_weights"), sizeof(float) * dm.R * nconv, hipMemcpyDeviceToDevice, s));
      COPER_HIP_TRY(h, hipMemcpyAsync(h->conv_b_rel, P("conv1_bias"), sizeof(float) * dm.R * dm.C, hipMemcpyDeviceToDevice, s));
    } else {
      const float* ctx; int K; char buf[256];
      if ((rc = run_generator_hidden(h, "conv1_weights", cfg.n_ctx_conv, cfg.ctx_conv, &ctx, &K, s))) return rc;
      snprintf(buf, sizeof buf, "conv1_weights/CPG/Projection%d", cfg.n_ctx_conv);
      if ((rc = launch_gen_small(h, ctx, dm.R, K, P(buf), nconv, nullptr, nullptr, false, h->conv_w_rel, s))) return rc;
      if ((rc = run_generator_hidden(h, "conv1_bias", cfg.n_ctx_conv, cfg.ctx_conv, &ctx, &K, s))) return rc;
      snprintf(buf, sizeof buf, "conv1_bias/CPG/Projection%d", cfg.n_ctx_conv);
      if ((rc = launch_gen_small(h, ctx, dm.R, K, P(buf), dm.C, nullptr, nullptr, false, h->conv_b_rel, s))) return rc;
    }
  }
  int64_t ksteps = dm.F_pad / 16;
  size_t per_rel = (size_t)dm.nfb * ksteps * 64 * 4;  // floats
  // coper_config.rel_mod_*: weight sets of the relations r with r % G == g only, relation r at slot r / G (ceil(R / G) slots on
  // every rank: a relation id the handle does not hold still indexes inside the planes; it is counted, kernels_dense_fused_bf16.hip)
  const int w_div = cfg.rel_mod_world > 1 ? cfg.rel_mod_world : 1, w_rem = w_div > 1 ? cfg.rel_mod_rank : 0;
  if (w_div > 1 && (!dm.gen_fc || dm.lookup || cfg.score_mode == COPER_SCORE_F32))
    return fail(h, COPER_EUNSUPPORTED, "rel_mod_world > 1: COPER_SCORE_BF16X3 with generated dense weights (context_rel_out) only");
  h->w_div = w_div; h->w_rem = w_rem;
  const int64_t n_own = w_div > 1 ? (dm.R > w_rem ? (dm.R - w_rem + w_div - 1) / w_div : 0) : dm.R;
  h->Rw = dm.gen_fc ? (w_div > 1 ? (dm.R + w_div - 1) / w_div : dm.R) : 1;
  if ((rc = dev_alloc(h, &h->Wf, per_rel * h->Rw))) return rc;
  float* ctx_sel = nullptr;       // the generator contexts of the held relations, compacted (freed behind the synchronize below)
  if (dm.gen_fc) {
    if ((rc = dev_alloc(h, &h->fc_b_rel, (size_t)dm.R * dm.d))) return rc;
    if (dm.lookup) {
      COPER_HIP_TRY(h, hipMemcpyAsync(h->fc_b_rel, P("fc_bias"), sizeof(float) * dm.R * dm.d, hipMemcpyDeviceToDevice, s));
      if ((rc = launch_gen_dense_frag(h, nullptr, dm.R, 0, P("fc_weights"), 1, h->Wf, s))) return rc;
    } else {
      const float* ctx; int K; char buf[256];
      if ((rc = run_generator_hidden(h, "fc_bias", cfg.n_ctx_out, cfg.ctx_out, &ctx, &K, s))) return rc;
      snprintf(buf, sizeof buf, "fc_bias/CPG/Projection%d", cfg.n_ctx_out);
      if ((rc = launch_gen_small(h, ctx, dm.R, K, P(buf), dm.d, nullptr, nullptr, false, h->fc_b_rel, s))) return rc;
      if ((rc = run_generator_hidden(h, "fc_weights", cfg.n_ctx_out, cfg.ctx_out, &ctx, &K, s))) return rc;
      snprintf(buf, sizeof buf, "fc_weights/CPG/Projection%d", cfg.n_ctx_out);
      if (w_div > 1) {      // slot i = relation w_rem + i w_div: the same context row, the same sum -- W_r is the same bits at any G
        if ((rc = dev_alloc(h, &ctx_sel, (size_t)h->Rw * K))) return rc;
        COPER_HIP_TRY(h, hipMemsetAsync(ctx_sel, 0, sizeof(float) * (size_t)h->Rw * K, s));
        if (n_own > 0)
          COPER_HIP_TRY(h, hipMemcpy2DAsync(ctx_sel, sizeof(float) * K, ctx + (size_t)w_rem * K, sizeof(float) * (size_t)w_div * K, sizeof(float) * K,
                                            (size_t)n_own, hipMemcpyDeviceToDevice, s));
        ctx = ctx_sel;
      }
      if ((rc = launch_gen_dense_frag(h, ctx, h->Rw, K, P(buf), 0, h->Wf, s))) return rc;
    }
  } else {
    if ((rc = launch_gen_dense_frag(h, nullptr, 1, 0, P("fc_weights"), 1, h->Wf, s))) return rc;
  }
  h->enc_bf16 = cfg.score_mode != COPER_SCORE_F32 && conv_bf16_supported(dm);
  if (w_div > 1 && !h->enc_bf16) {
    dev_free(&ctx_sel);
    return fail(h, COPER_EUNSUPPORTED, "rel_mod_world > 1: the configuration is not served by the 16-bit encoder");
  }
  if (h->enc_bf16) {
    size_t plane = (size_t)h->Rw * dm.nfb * w16_ks_stride(dm) * 64 * 16;
    dev_free((char**)&h->Wf16_hi); dev_free((char**)&h->Wf16_lo);
    if (tracked_malloc(&h->Wf16_hi, plane) != hipSuccess || tracked_malloc(&h->Wf16_lo, plane) != hipSuccess)
      return fail(h, COPER_ENOMEM, "hipMalloc of the bf16 weight planes failed");
    // powers of two of the encoder's operands (split16.h): e_W per relation from its own largest |W|, e_x from a bound on x
    if (w_div > 1) {       // e_W is looked up by RELATION ID everywhere; the conversion works on slots: computed there, scattered to the ids
      int32_t* by_slot = nullptr;
      if ((rc = dev_alloc(h, &h->w_exp, (size_t)dm.R)) || (rc = dev_alloc(h, &by_slot, (size_t)h->Rw))) return rc;
      COPER_HIP_TRY(h, hipMemsetAsync(h->w_exp, 0, sizeof(int32_t) * (size_t)dm.R, s));
      int32_t* by_id = h->w_exp;
      h->w_exp = by_slot;
      rc = launch_wfrag_to_bf16(h, h->Wf, h->Rw, h->Wf16_hi, h->Wf16_lo, s);
      h->w_exp = by_id;
      if (!rc && n_own > 0 &&
          hipMemcpy2DAsync(by_id + w_rem, sizeof(int32_t) * (size_t)w_div, by_slot, sizeof(int32_t), sizeof(int32_t), (size_t)n_own,
                           hipMemcpyDeviceToDevice, s) != hipSuccess)
        rc = fail(h, COPER_EHIP, "hipMemcpy2DAsync (e_W by relation id)");
      hipError_t e = hipStreamSynchronize(s);
      dev_free(&by_slot);
      if (rc) return rc;
      COPER_HIP_TRY(h, e);
    } else {
      if ((rc = dev_alloc(h, &h->w_exp, (size_t)h->Rw))) return rc;
      if ((rc = launch_wfrag_to_bf16(h, h->Wf, h->Rw, h->Wf16_hi, h->Wf16_lo, s))) return rc;
    }
    if ((rc = compute_x_exp(h, h->band_consts + 5, s))) return rc;
    COPER_HIP_TRY(h, hipStreamSynchronize(s));
    dev_free(&h->Wf);  // the fp32 image was only the staging form
    dev_free(&ctx_sel);
  }
  }                    // ---- (role_enc)
  // entity table image(s) (a COPER_ROLE_ENCODE handle has none)
  if (!role_score) {
  } else if (cfg.score_mode == COPER_SCORE_F32) {
    if ((rc = dev_alloc(h, &h->Ef, (size_t)dm.n_eblk * dm.KS * 64 * 4)) || (rc = dev_alloc(h, &h->bias_pad, (size_t)dm.n_eblk * 32)))
      return rc;
    if ((rc = launch_entity_frag(h, P("ent_emb"), P("pred_bias"), s))) return rc;
  } else {
    // bias_pad comes from the (tiny-KS) fp32 image builder run on a 1-k-step view; the table goes to two bf16 planes
    size_t plane = (size_t)dm.n_eblk * dm.KS16 * 64 * 16;
    dev_free((char**)&h->Ef16_hi); dev_free((char**)&h->Ef16_lo);
    dev_free((char**)&h->Erm16_hi); dev_free((char**)&h->Erm16_lo);
    if (tracked_malloc(&h->Ef16_hi, plane) != hipSuccess || tracked_malloc(&h->Ef16_lo, plane) != hipSuccess ||
        tracked_malloc(&h->Erm16_hi, plane) != hipSuccess || tracked_malloc(&h->Erm16_lo, plane) != hipSuccess)
      return fail(h, COPER_ENOMEM, "hipMalloc of the bf16 entity planes failed");
    if ((rc = dev_alloc(h, &h->bias_pad, (size_t)dm.n_eblk * 32))) return rc;
    if ((rc = launch_bias_pad(h, P("pred_bias"), s))) return rc;
    const size_t f3 = (size_t)dm.n_eblk * 2 * f3_steps(dm.KS16) * 2 * 64 * 16;    // the count kernel's image (bf16x3_chain.h)
    dev_free((char**)&h->Ef3);
    if (tracked_malloc(&h->Ef3, f3) != hipSuccess) { (void)hipGetLastError(); return fail(h, COPER_ENOMEM, "hipMalloc of the entity image failed"); }
    COPER_HIP_TRY(h, hipMemsetAsync(h->Ef3, 0, f3, s));
    if ((rc = launch_rows_to_frag_bf16(h, P("ent_emb"), dm.n_local, dm.n_eblk, (uint4*)h->Ef16_hi, (uint4*)h->Ef16_lo,
                                       (uint4*)h->Erm16_hi, (uint4*)h->Erm16_lo, (uint4*)h->Ef3, false, s)))
      return rc;
    if ((rc = score_bf16_kernels_init(h))) return rc;
  }
  {
    hipDeviceProp_t prop;
    COPER_HIP_TRY(h, hipGetDeviceProperties(&prop, cfg.device));
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int64_t share = (int64_t)(prop.totalGlobalMem / 32 / sizeof(float));
    h->gmax_max_floats = share > ((int64_t)1 << 28) ? share : ((int64_t)1 << 28);
  }
  if (role_score && cfg.score_mode == COPER_SCORE_F32 && (rc = score_kernels_init(h))) return rc;
  if ((rc = fused_fin_update(h, s))) return rc;       // (parameters / exponents moved)
  h->prepared = true;
  return COPER_OK;
}

COPER_API int coper_reserve(coper_handle* h, int64_t max_queries, int64_t max_filter_nnz, void* stream) {
  if (!h || max_queries < 0 || max_filter_nnz < 0) return fail(h, COPER_EINVAL, "coper_reserve: bad argument");
  COPER_HIP_TRY(h, hipSetDevice(h->cfg.device));
  return ensure_rank_workspace(h, max_queries, max_filter_nnz, true, (hipStream_t)stream);
}

#define COPER_REQUIRE_PREPARED(h)                                                           \
  do {                                                                                      \
    if (!(h)) return COPER_EINVAL;                                                          \
    if (!(h)->prepared) return fail((h), COPER_ESTATE, "coper_prepare has not been called"); \
  } while (0)
// coper_config.role: the entry points of the other role are refused, loudly (their derived buffers do not exist)
#define COPER_REQUIRE_ENCODER(h)                                                                                                   \
  do {                                                                                                                             \
    if ((h)->cfg.role == COPER_ROLE_SCORE) return fail((h), COPER_ESTATE, "the handle was created with COPER_ROLE_SCORE: no encoder"); \
  } while (0)
#define COPER_REQUIRE_SCORER(h)                                                                                                     \
  do {                                                                                                                              \
    if ((h)->cfg.role == COPER_ROLE_ENCODE) return fail((h), COPER_ESTATE, "the handle was created with COPER_ROLE_ENCODE: no scorer"); \
  } while (0)

COPER_API int coper_copy_out_i32(coper_handle* h, const int32_t* src, int64_t n, int32_t* dst, void* stream) {
  if (!h) return COPER_EINVAL;
  if (n == 0) return COPER_OK;
  if (!src || !dst || n < 0) return fail(h, COPER_EINVAL, "coper_copy_out_i32: bad argument");
  return launch_copy_i32(h, src, n, dst, (hipStream_t)stream);
}

COPER_API int coper_widen_ids(coper_handle* h, const int32_t* src, int64_t n, int64_t* dst, void* stream) {
  if (!h) return COPER_EINVAL;
  if (n == 0) return COPER_OK;
  if (!src || !dst || n < 0) return fail(h, COPER_EINVAL, "coper_widen_ids: bad argument");
  return launch_widen_ids(h, src, n, dst, (hipStream_t)stream);
}

COPER_API int coper_stage_ids_next(coper_handle* h, const int32_t* src, int64_t n, int64_t* dst) {
  if (!h) return COPER_EINVAL;
  if (n < 0 || (n > 0 && (!src || !dst))) return fail(h, COPER_EINVAL, "coper_stage_ids_next: bad argument");
  h->pipe.stage.src = src; h->pipe.stage.n = n; h->pipe.stage.dst = dst;
  return COPER_OK;
}

COPER_API int coper_post_i32_next(coper_handle* h, const int32_t* src, int64_t n, int32_t* dst) {
  if (!h) return COPER_EINVAL;
  if (n < 0 || (n > 0 && (!src || !dst))) return fail(h, COPER_EINVAL, "coper_post_i32_next: bad argument");
  h->pipe.post.src = src; h->pipe.post.n = n; h->pipe.post.dst = dst;
  return COPER_OK;
}

COPER_API int coper_group_next(coper_handle* h, const int64_t* e1, const int64_t* rel, int64_t B, int32_t have_e1_rows) {
  if (!h) return COPER_EINVAL;
  if (B < 0 || (B > 0 && (!rel || (!e1 && !have_e1_rows)))) return fail(h, COPER_EINVAL, "coper_group_next: bad argument");
  auto& g = h->pipe.gnext;
  g.e1 = have_e1_rows ? nullptr : e1; g.rel = rel; g.B = B; g.rows = have_e1_rows ? 1 : 0;
  g.pending = B > 0;
  g.ride = false;
  return COPER_OK;
}

COPER_API int coper_gather_entities(coper_handle* h, const int64_t* ids, int64_t B, float* out, void* stream) {
  COPER_REQUIRE_PREPARED(h);
  if (B == 0) return COPER_OK;
  if (!ids || !out || B < 0) return fail(h, COPER_EINVAL, "coper_gather_entities: bad argument");
  return launch_gather_entities(h, ids, B, out, (hipStream_t)stream);
}

static bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return true; }
  return cap != hipStreamCaptureStatusNone;
}

// grouping, conv and the dense layer up to the K-slice partials in z_part (everything of coper_encode but the finalize)
// h_x3: where the x3 encoder may write finished h rows when its launch can finalize them itself (dense_fused_finalizes);
// *finalized says whether it did (the caller then skips its finalize launch)
static int encode_partials(coper_handle* h, const int64_t* e1, const int64_t* rel, int64_t B, const float* e1_rows, hipStream_t s,
                           int* ksplit_out, float* h_out_f32_path, float* h_x3 = nullptr, bool* finalized = nullptr,
                           bool consume_prepared = false) {
  if (finalized) *finalized = false;
  const Dims& dm = h->dm;
  int rc;
  if ((rc = ensure_workspace(h, B, 0, s))) return rc;
  const int tq = 32;
  // K-split of the dense layer depends on F only, never on the batch: h[b] is then a pure function of
  // (e1[b], rel[b]) -- bit-identical whatever batch, chunking or rank computes it.
  int64_t ksteps = dm.F_pad / 16;
  // Number of K slices: a function of the CONFIGURATION only (never of the batch).  More slices = more workgroups
  // per weight stream (fills the chip when there are few streams) but more partial-sum traffic and prologues;
  // measured on MI355X (same box, bf16x3): 474 relations 1 > 2 > 3 > 4 > 8 (one slice = one workgroup per relation tile,
  // 237 of them on 256 CUs: fused encoder 0.184 vs 0.189 ms, and half the partial sums for the tail kernel to read: pass
  // 0.526 vs 0.534 ms); 2,000 relations (the 10M-entity config, ~1,000 tiles of ~4 queries) 2 > 1 (1.00 vs 1.30 ms);
  // 22 relations 8 > 4 > 3 > 2; one shared weight (plain ConvE) 3 > 4 > 8 > 2.
  int ksplit = 1;
  // (the fp32 encoder -- conv kernel, x through HBM, k_dense_big_f32 -- keeps two slices there: 0.455 vs 0.515 ms)
  const int mid = h->enc_bf16 ? 1 : 2;
  if (ksteps >= 64) ksplit = !dm.gen_fc ? 3 : dm.R >= 1024 ? 2 : dm.R >= 256 ? mid : dm.R >= 32 ? 4 : 8;
  if (ksplit > h->ws_ksplit) ksplit = h->ws_ksplit;
  *ksplit_out = ksplit;
  // coper_group_next: a pass whose batch was sorted in the shadow of the last one skips its grouping launches.  Only a pass that
  // writes ranks (consume_prepared: coper_encode_rank) may run on such a grouping: its tiles check it against the live ids and the
  // kernel that presets its rank counters acts on the verdict (group_body.h: the guard).  Whatever else comes next drops it.
  bool pre = false;
  {
    int set = 0;
    pre = h->pipe.take_prepared(e1, rel, B, e1_rows ? 1 : 0, &set) && consume_prepared && h->gset[set].slab != nullptr && !stream_is_capturing(s);
    group_use_set(h, pre ? set : 0);
    h->pass_chk = pre ? h->gset[set].chk : nullptr;
  }
  h->pipe.post.here = false;
  if (!pre) {
    ScopedKernelTimer t(h, "group", s);
    if ((rc = launch_group_by_relation(h, e1, rel, e1_rows != nullptr, B, tq, s))) return rc;
  }
  if (h->enc_bf16) {
    const bool fused = dense_fused_supported(h, ksplit);
    // a coper_group_next registration is for THIS call only: it rides in the fused launch below or is dropped (a captured pass, a
    // configuration without the fused encoder) -- never kept for a later call, whose caller may have freed the arrays it names
    const bool want_group = h->pipe.gnext.pending;
    h->pipe.gnext.pending = false;
    if (fused && want_group && !stream_is_capturing(s)) {
      if ((rc = group_sets_ensure(h, s))) return rc;
      h->pipe.gnext.ride = true;
    }
    if (pre && h->pipe.post.n > 0) {       // the job that rides in the grouping launch otherwise
      if (fused) h->pipe.post.here = true;
      else if (const int64_t pn = h->pipe.take_post()) {
        if ((rc = launch_copy_i32(h, h->pipe.post.src, pn, h->pipe.post.dst, s))) return rc;
      }
    }
    if (!fused) {                           // (a pending coper_stage_ids_next rides in the fused launch only)
      if (const int64_t n = h->pipe.take_stage())
        if ((rc = launch_widen_ids(h, h->pipe.stage.src, n, h->pipe.stage.dst, s))) return rc;
    }
    if (!fused && h->w_div > 1)
      return fail(h, COPER_EUNSUPPORTED, "rel_mod_world > 1: the configuration is not served by the fused encoder");
    if (fused) {   // one launch serves every tile: conv, BN, ReLU and the dense layer (kernels_dense_fused_bf16.hip)
      ScopedKernelTimer t(h, "dense", s);
      if (finalized) *finalized = dense_fused_finalizes(h, ksplit, h_x3);
      return launch_dense_fused_bf16(h, e1, rel, e1_rows, B, ksplit, h_x3, s);
    }
    if ((rc = launch_conv_bf16(h, e1, rel, e1_rows, B, false, s))) return rc;
    {
      ScopedKernelTimer t(h, "dense", s);
      if ((rc = launch_dense_bf16(h, B, ksplit, false, s))) return rc;
    }
    return COPER_OK;
  }
  h->pipe.gnext.pending = false;
  if (const int64_t n = h->pipe.take_stage())
    if ((rc = launch_widen_ids(h, h->pipe.stage.src, n, h->pipe.stage.dst, s))) return rc;
  if ((rc = launch_conv(h, e1, rel, e1_rows, B, s))) return rc;
  return launch_dense(h, rel, B, tq, ksplit, h_out_f32_path, s);   // fp32 mode: finalize included
}

COPER_API int coper_encode(coper_handle* h, const int64_t* e1, const int64_t* rel, int64_t B, const float* e1_rows,
                 float* h_out, void* stream) {
  COPER_REQUIRE_PREPARED(h);
  COPER_REQUIRE_ENCODER(h);
  if (B == 0) return COPER_OK;  // empty batch: nothing to do (the reference's session.run on an empty batch returns [0,d])
  if (!rel || !h_out || B < 0 || (!e1 && !e1_rows)) return fail(h, COPER_EINVAL, "coper_encode: bad argument");
  if (B > 0x7fffffff) return fail(h, COPER_EINVAL, "coper_encode: batch too large");
  hipStream_t s = (hipStream_t)stream;
  int rc, ksplit = 1;
  bool finalized = false;
  if ((rc = encode_partials(h, e1, rel, B, e1_rows, s, &ksplit, h_out, h_out, &finalized))) return rc;
  if (h->enc_bf16 && !finalized) return launch_dense_finalize(h, rel, B, ksplit, h_out, s);
  return COPER_OK;
}

COPER_API int coper_score_all(coper_handle* h, const float* hvec, int64_t B, float* logits, int64_t ld, void* stream) {
  COPER_REQUIRE_PREPARED(h);
  COPER_REQUIRE_SCORER(h);
  if (B == 0) return COPER_OK;
  if (!hvec || !logits || B < 0 || ld < h->dm.n_local) return fail(h, COPER_EINVAL, "coper_score_all: bad argument");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = ensure_workspace(h, B, 0, s))) return rc;
  return score_all_dispatch(h, hvec, B, logits, ld, s);
}

COPER_API int coper_score_lookup(coper_handle* h, const float* hvec, const int32_t* lookup, int64_t B, int64_t L, float* out,
                       void* stream) {
  COPER_REQUIRE_PREPARED(h);
  COPER_REQUIRE_SCORER(h);
  if (B == 0 || L == 0) return COPER_OK;  // eval batches carry lookup_values of shape [B, 0] (data.py:205-213)
  if (!hvec || !lookup || !out || B < 0 || L < 0) return fail(h, COPER_EINVAL, "coper_score_lookup: bad argument");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = ensure_workspace(h, B, 0, s))) return rc;
  if (h->cfg.score_mode != COPER_SCORE_F32) return launch_score_lookup_bf16x3(h, hvec, lookup, B, L, out, s);
  return launch_score_lookup(h, hvec, lookup, B, L, out, s);
}

COPER_API int coper_target_scores(coper_handle* h, const float* hvec, const int64_t* e2, int64_t B, float* tgt, void* stream) {
  COPER_REQUIRE_PREPARED(h);
  COPER_REQUIRE_SCORER(h);
  if (B == 0) return COPER_OK;
  if (!hvec || !e2 || !tgt || B < 0) return fail(h, COPER_EINVAL, "coper_target_scores: bad argument");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = ensure_workspace(h, B, 0, s))) return rc;
  if (h->cfg.score_mode != COPER_SCORE_F32) {
    if ((rc = launch_pair_targets_bf16x3(h, hvec, e2, B, tgt, s))) return rc;
    return launch_exact_targets(h, hvec, e2, B, tgt + B, s);     // the fp32-chain logit: what the exact band compares against
  }
  if ((rc = launch_pair_targets(h, hvec, e2, B, tgt, s))) return rc;
  COPER_HIP_TRY(h, hipMemcpyAsync(tgt + B, tgt, sizeof(float) * B, hipMemcpyDeviceToDevice, s));   // the mode's logit IS the chain's
  return COPER_OK;
}

COPER_API int coper_score_rows(coper_handle* h, const float* hvec, const float* rows, const float* bias, int64_t B, float* out,
                               void* stream) {
  COPER_REQUIRE_PREPARED(h);
  if (B == 0) return COPER_OK;
  if (!hvec || !rows || !bias || !out || B < 0) return fail(h, COPER_EINVAL, "coper_score_rows: bad argument");
  return launch_exact_rows(h, hvec, rows, bias, B, out, (hipStream_t)stream);     // (the chain is the f32 mode's own logit)
}

COPER_API int coper_rank_counts(coper_handle* h, const float* hvec, const float* tgt, const int64_t* e2,
                      const int64_t* filt_indptr, const int64_t* filt_idx, int64_t filt_nnz, int64_t B, int32_t k,
                      int32_t* n_greater, int32_t* n_equal, float* topk_val, int64_t* topk_idx, void* stream) {
  COPER_REQUIRE_PREPARED(h);
  COPER_REQUIRE_SCORER(h);
  if (B == 0) return COPER_OK;
  if (!hvec || !tgt || !e2 || !filt_indptr || !n_greater || B < 0 || filt_nnz < 0 ||
      (filt_nnz > 0 && !filt_idx))
    return fail(h, COPER_EINVAL, "coper_rank_counts: bad argument");
  if (k < 0 || k > 1024 || (k > 0 && (!topk_val || !topk_idx)))
    return fail(h, COPER_EINVAL, "coper_rank_counts: bad top-k arguments (0 <= k <= 1024)");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = ensure_rank_workspace(h, B, filt_nnz, false, s))) return rc;
  // 0 < k <= COPER_TOPK_PRUNED_MAX (128): the count pass also writes block maxima and the top-k is selected from the few blocks that can
  // hold it (kernels_topk_bf16.hip): no logits workspace
  const int XF = topk_expand(h);      // (2 on large tables: 64-entity candidate blocks, expanded to two 32-entity ones)
  const bool pruned = k > 0 && k <= COPER_TOPK_PRUNED_MAX &&
                      XF * ((int64_t)k * B + filt_nnz) + 32 * h->dm.n_eblk * topk_nseg(h->dm.n_eblk) < 0x7fffffffLL;   // int32 slot ids
  if (pruned) {
    const size_t gneed = (size_t)(topk_gm_rows(h) * topk_chunk_queries(h->dm.n_eblk, B, h->gmax_max_floats));
    const size_t t64 = (size_t)((int64_t)k * B + filt_nnz), tneed = (size_t)XF * t64;
    const size_t cneed = topk_coarse_bytes(topk_gm_rows(h), topk_chunk_queries(h->dm.n_eblk, B, h->gmax_max_floats));
    if (gneed > h->gmax_cap || tneed > h->cand_cap || (size_t)B > h->cand_tau_cap || cneed > h->tk_coarse_cap) {
      COPER_HIP_TRY(h, hipStreamSynchronize(s));
      if (cneed > h->tk_coarse_cap) {
        if ((rc = dev_alloc(h, &h->tk_coarse_ws, cneed))) return rc;
        h->tk_coarse_cap = cneed;
      }
      if (gneed > h->gmax_cap) {
        if ((rc = dev_alloc(h, &h->gmax_ws, gneed))) return rc;
        h->gmax_cap = gneed;
      }
      if (tneed > h->cand_cap) {
        h->cand_cap = 0;
        const size_t tlist = tneed + (XF > 1 ? t64 : 0);      // (+ the 64-entity level's own lists, behind the expanded ones)
        if ((rc = dev_alloc(h, &h->cand_blk_ws, tlist)) || (rc = dev_alloc(h, &h->cand_q_ws, tlist)) ||
            (rc = dev_alloc(h, &h->cand_val_ws, tneed * 32)) ||
            (rc = dev_alloc(h, &h->cand_sorted_ws, topk_sorted_cap(h->dm.n_eblk * topk_nseg(h->dm.n_eblk), (int64_t)tneed))))
          return rc;
        h->cand_cap = tneed;
      }
      if ((size_t)B > h->cand_tau_cap) {
        if ((rc = dev_alloc(h, &h->cand_tau_ws, (size_t)B))) return rc;
        h->cand_tau_cap = (size_t)B;
      }
      if (!h->blk_cnt_ws) {
        const size_t gv = (size_t)(h->dm.n_eblk * topk_nseg(h->dm.n_eblk));
        if ((rc = dev_alloc(h, &h->blk_cnt_ws, 2 * gv)) || (rc = dev_alloc(h, &h->blk_off_ws, gv + 1 + gv / 4096 + 2))) return rc;   // + chunk sums of the scan
      }
    }
  }
  if (h->cfg.score_mode != COPER_SCORE_F32) {
    if (!(h->trust_packed && h->packed_hvec == hvec && h->packed_B == B) && (rc = launch_pack_h_bf16(h, hvec, B, s))) return rc;
    COPER_DBG_SYNC(h, s, "pack_h");
    // the exact band of every query around the mode's target logit; comparisons inside it are decided against tgt[B ..]
    if ((rc = launch_band_setup(h, hvec, tgt, B, s))) return rc;
    COPER_DBG_SYNC(h, s, "band_setup");
    if (pruned)
      rc = launch_topk_pruned_bf16x3(h, hvec, tgt + B, e2, filt_indptr, filt_idx, filt_nnz, B, k, n_greater, n_equal, topk_val, topk_idx, s);
    else
      rc = launch_score_count_bf16x3(h, hvec, tgt + B, e2, filt_indptr, filt_idx, B, n_greater, n_equal, s);
    if (rc) return rc;
    if ((rc = launch_filter_correct_bf16x3(h, e2, filt_indptr, filt_idx, filt_nnz, B, n_greater, s))) return rc;
    COPER_DBG_SYNC(h, s, "filter_correct");
  } else {
    if (pruned)
      rc = launch_topk_pruned_f32(h, hvec, tgt, e2, filt_indptr, filt_idx, filt_nnz, B, k, n_greater, n_equal, topk_val, topk_idx, s);
    else
      rc = launch_score_count(h, hvec, tgt, B, n_greater, n_equal, s);
    if (rc) return rc;
    if ((rc = launch_filter_correct(h, hvec, tgt, e2, filt_indptr, filt_idx, filt_nnz, B, n_greater, n_equal, s))) return rc;
  }
  if (pruned) return COPER_OK;
  if (k > 0) {
    // logits workspace: at most 256 MiB (or one row) at a time
    int64_t rows = (int64_t)(256ll << 20) / (h->dm.n_local * 4);
    if (rows < 1) rows = 1;
    if (rows > B) rows = B;
    if (rows > h->logits_ws_rows) {
      COPER_HIP_TRY(h, hipStreamSynchronize(s));
      if ((rc = dev_alloc(h, &h->logits_ws, (size_t)rows * h->dm.n_local))) return rc;
      h->logits_ws_rows = rows;
    }
    return launch_topk(h, hvec, e2, filt_indptr, filt_idx, B, k, topk_val, topk_idx, h->logits_ws, h->logits_ws_rows, s);
  }
  return COPER_OK;
}

COPER_API int coper_rank(coper_handle* h, const float* hvec, const int64_t* e2, const int64_t* filt_indptr,
               const int64_t* filt_idx, int64_t filt_nnz, int64_t B, int32_t* ranks, int32_t* n_equal, void* stream) {
  COPER_REQUIRE_PREPARED(h);
  COPER_REQUIRE_SCORER(h);
  if (B == 0) return COPER_OK;
  if (!ranks || B < 0) return fail(h, COPER_EINVAL, "coper_rank: bad argument");
  if (h->dm.n_local != h->dm.E) return fail(h, COPER_ESTATE, "coper_rank needs the whole table; sharded handles use coper_target_scores + coper_rank_counts");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if ((rc = ensure_rank_workspace(h, B, filt_nnz, false, s))) return rc;
  const bool direct = h->cfg.score_mode != COPER_SCORE_F32;
  if (direct) {
    // bf16x3: n_greater accumulates straight into `ranks` started from 1 (no finish launch), preset by the packing
    // launch of the target pass (no zeroing launch)
    h->count_base = 1;
    h->preset_cnt = ranks;
    h->preset_eq = n_equal;
  }
  if (direct) h->expand_indptr = filt_indptr;   // the target pass expands the CSR rows for the filter pass
  rc = coper_target_scores(h, hvec, e2, B, h->tgt_ws, stream);
  h->expand_indptr = nullptr;
  h->preset_cnt = nullptr;
  h->preset_eq = nullptr;
  if (rc) { h->count_base = 0; h->counts_preset = nullptr; return rc; }
  int32_t* ng = direct ? ranks : h->cnt_ws;
  int32_t* ne = n_equal;  // NULL: ties are not counted
  h->trust_packed = true;  // same hvec, same stream, no caller code in between: the packing of target_scores is valid
  rc = coper_rank_counts(h, hvec, h->tgt_ws, e2, filt_indptr, filt_idx, filt_nnz, B, 0, ng, ne, nullptr, nullptr, stream);
  h->trust_packed = false;
  h->count_base = 0;
  h->counts_preset = nullptr;
  if (rc) return rc;
  if (direct) return COPER_OK;
  return launch_finish_ranks(h, ng, B, ranks, s);
}

COPER_API int coper_encode_rank(coper_handle* h, const int64_t* e1, const int64_t* rel, const float* e1_rows, const int64_t* e2,
                                const int64_t* filt_indptr, const int64_t* filt_idx, int64_t filt_nnz, int64_t B, float* h_out,
                                int32_t* ranks, int32_t* n_equal, void* stream) {
  COPER_REQUIRE_PREPARED(h);
  COPER_REQUIRE_ENCODER(h);
  COPER_REQUIRE_SCORER(h);
  if (B == 0) return COPER_OK;
  if (!rel || !e2 || !filt_indptr || !ranks || B < 0 || filt_nnz < 0 || (!e1 && !e1_rows) || (filt_nnz > 0 && !filt_idx))
    return fail(h, COPER_EINVAL, "coper_encode_rank: bad argument");
  if (B > 0x7fffffff) return fail(h, COPER_EINVAL, "coper_encode_rank: batch too large");
  if (h->dm.n_local != h->dm.E) return fail(h, COPER_ESTATE, "coper_encode_rank needs the whole table (see coper_rank)");
  hipStream_t s = (hipStream_t)stream;
  int rc, ksplit = 1;
  struct ClearChk { coper_handle* h; ~ClearChk() { h->pass_chk = nullptr; } } clear_chk{h};   // (the guard's verdict belongs to this call's launches only)
  // the embedding is needed in fp32 either way: the fp32-exact mode scores from it, the bf16x3 mode's exact band re-scores from it
  if ((rc = ensure_rank_workspace(h, B, filt_nnz, h_out == nullptr, s))) return rc;
  float* hv = h_out ? h_out : h->h_ws;
  if (!h->enc_bf16) {
    // fp32-exact mode (or a configuration the bf16x3 encoder does not serve): the two-call path
    if ((rc = coper_encode(h, e1, rel, B, e1_rows, hv, stream))) return rc;
    return coper_rank(h, hv, e2, filt_indptr, filt_idx, filt_nnz, B, ranks, n_equal, stream);
  }
  // bf16x3: the finalize writes h straight into the planes the rank kernels read (and the fp32 rows the exact band needs) and
  // presets the counters, which accumulate into `ranks` from 1: no pack, zero or finish launch
  bool finalized = false;
  if ((rc = encode_partials(h, e1, rel, B, e1_rows, s, &ksplit, nullptr, hv, &finalized, true))) return rc;
  if (!n_equal && tail_fused_supported(h)) {
    // ranks only (what the reference computes): finalize, targets, the band and the filter correction in ONE launch
    // (kernels_tail_bf16.hip) that leaves ranks = 1 - (known answers above the band); the count kernel and the exact
    // decision of the band add to it
    {
      ScopedKernelTimer t(h, "tail", s);
      if ((rc = launch_finalize_targets_filter_bf16x3(h, B, finalized ? 0 : ksplit, hv, e2, filt_indptr, filt_idx, filt_nnz, h->tgt_ws, ranks, s))) return rc;
    }
    h->counts_preset = ranks;
    h->count_base = 1;
    rc = launch_score_count_bf16x3(h, hv, nullptr, e2, filt_indptr, filt_idx, B, ranks, nullptr, s);
    h->count_base = 0;
    h->counts_preset = nullptr;
    h->excess_pending = false;     // (consumed by the first band launch; a failed launch must not leave it to a later pass)
    return rc;
  }
  if ((rc = launch_dense_finalize_pack(h, B, finalized ? 0 : ksplit, hv, ranks, 1, n_equal, s))) return rc;
  h->expand_indptr = filt_indptr;
  rc = launch_pair_targets_packed_bf16x3(h, e2, B, h->tgt_ws, s);
  h->expand_indptr = nullptr;
  if (rc) return rc;
  if ((rc = launch_exact_targets(h, hv, e2, B, h->tgt_ws + B, s))) return rc;
  h->packed_hvec = hv;
  h->packed_B = B;
  h->trust_packed = true;
  h->count_base = 1;
  h->counts_preset = ranks;
  rc = coper_rank_counts(h, hv, h->tgt_ws, e2, filt_indptr, filt_idx, filt_nnz, B, 0, ranks, n_equal, nullptr, nullptr, stream);
  h->trust_packed = false;
  h->count_base = 0;
  h->counts_preset = nullptr;
  return rc;
}

COPER_API int coper_band_audit(coper_handle* h, int32_t reset, float* max_ratio, int64_t* n_pairs, void* stream) {
  if (!h) return COPER_EINVAL;
  if (max_ratio) *max_ratio = 0.f;
  if (n_pairs) *n_pairs = 0;
  if (!h->band_consts) return COPER_OK;      // fp32-exact mode: no band
  hipStream_t s = (hipStream_t)stream;
  unsigned v[2] = {0u, 0u};
  COPER_HIP_TRY(h, hipMemcpyAsync(v, h->band_consts + 3, sizeof v, hipMemcpyDeviceToHost, s));
  if (reset) COPER_HIP_TRY(h, hipMemsetAsync(h->band_consts + 3, 0, sizeof v, s));
  COPER_HIP_TRY(h, hipStreamSynchronize(s));
  if (max_ratio) memcpy(max_ratio, &v[0], sizeof(float));
  if (n_pairs) *n_pairs = (int64_t)v[1];
  return COPER_OK;
}

COPER_API int coper_band_audit_post(coper_handle* h, int32_t reset, uint32_t* dst2, void* stream) {
  if (!h || !dst2) return COPER_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (!h->band_consts) {       // fp32-exact mode: no band; zeros through the same kind of launch (dst2 may be host memory)
    return fail(h, COPER_ESTATE, "coper_band_audit_post: the handle has no band (COPER_SCORE_F32)");
  }
  int rc = launch_copy_i32(h, (const int32_t*)(h->band_consts + 3), 2, (int32_t*)dst2, s);
  if (rc) return rc;
  if (reset) COPER_HIP_TRY(h, hipMemsetAsync(h->band_consts + 3, 0, 2 * sizeof(unsigned), s));
  return COPER_OK;
}

COPER_API int coper_post_ranks_audit(coper_handle* h, const int32_t* ranks, int64_t n, int32_t* dst, int32_t reset, void* stream) {
  if (!h || n < 0 || !dst || (n > 0 && !ranks)) return fail(h, COPER_EINVAL, "coper_post_ranks_audit: bad argument");
  return launch_copy_i32_audit(h, ranks, n, dst, h->band_consts ? h->band_consts + 3 : nullptr, reset, (hipStream_t)stream);
}

// ---- step 1 of the entity-sharded exchange (coper_amd/sharding.py): the owned rows packed for the all-gather, the gathered rows handed out ----
COPER_API int coper_pack_owned_rows(coper_handle* h, const int64_t* local_rows, int64_t n, int64_t cap, float hdr0, float hdr1, float* buf,
                                    void* stream) {
  if (!h || n < 0 || cap < n || !buf || (n > 0 && !local_rows)) return fail(h, COPER_EINVAL, "coper_pack_owned_rows: bad argument");
  auto e = h->params.find("ent_emb"), b = h->params.find("pred_bias");
  if (e == h->params.end() || b == h->params.end() || !e->second.set || !b->second.set)
    return fail(h, COPER_EMISSING, "coper_pack_owned_rows: ent_emb / pred_bias were never set");
  return launch_pack_owned_rows(h, e->second.ptr, b->second.ptr, local_rows, n, cap, hdr0, hdr1, buf, (hipStream_t)stream);
}

COPER_API int coper_unpack_rows(coper_handle* h, const float* gathered, const int64_t* take1, const int64_t* take2, int64_t B, float* rows1,
                                float* rows2, float* bias2, void* stream) {
  if (!h || B < 0 || (B > 0 && (!gathered || !take1 || !take2 || !rows1 || !rows2 || !bias2))) return fail(h, COPER_EINVAL, "coper_unpack_rows: bad argument");
  if (B == 0) return COPER_OK;
  return launch_unpack_rows(h, gathered, take1, take2, B, rows1, rows2, bias2, (hipStream_t)stream);
}

// ---- step 3 of the entity-sharded exchange (coper_amd/sharding.py): the per-shard record packed, the gathered records merged ----
COPER_API int coper_pack_shard_record(coper_handle* h, const int32_t* n_greater, const int32_t* n_equal, const float* topk_val,
                                      const int64_t* topk_idx, int64_t B, int32_t k, int32_t reset_audit, int64_t* rec, void* stream) {
  if (!h || B < 0 || k < 0 || !rec || (B > 0 && (!n_greater || !n_equal)) || (B > 0 && k > 0 && (!topk_val || !topk_idx)))
    return fail(h, COPER_EINVAL, "coper_pack_shard_record: bad argument");
  return launch_pack_shard_record(h, n_greater, n_equal, topk_val, topk_idx, B, k, h->band_consts ? h->band_consts + 3 : nullptr, reset_audit, rec,
                                  (hipStream_t)stream);
}

COPER_API int coper_merge_shard_records(coper_handle* h, const int64_t* all_rec, int32_t world, int64_t B, int32_t k, int32_t* ranks,
                                        int32_t* n_equal, float* cand_val, int64_t* cand_idx, void* stream) {
  if (!h || world < 1 || B < 0 || k < 0 || (B > 0 && (!all_rec || !ranks)) || (B > 0 && k > 0 && (!cand_val || !cand_idx)))
    return fail(h, COPER_EINVAL, "coper_merge_shard_records: bad argument");
  if (B == 0) return COPER_OK;
  return launch_merge_shard_records(h, all_rec, world, B, k, ranks, n_equal, cand_val, cand_idx, (hipStream_t)stream);
}

// ---- host marshalling of a batch: int64 ids -> the int32 staging buffer, checked on the way ----
// One pass over an id array on the host: narrowed to int32 into dst (typically the pinned buffer coper_widen_ids reads) while the
// two things the staging path must know are checked -- every value fits int32 (status bit 0 otherwise), and, with a CSR indptr,
// every row of the array is ascending (bit 1 otherwise: the rank kernels' filter contract; data.py's canonical_csr is the
// slow path that sorts such rows).  Replaces three NumPy passes (canonical check, range check, converting copy) of
// ranking_and_hits' list route: 100 -> 25 us per 100 K filter entries.
// (the loops below are plain and flat so that the compiler vectorises them; the AVX2 clones are picked at run time -- 64-bit
//  compares do not exist in SSE2, and these passes are the host's share of a pass's critical path)
#define COPER_HOST_INLINE static inline __attribute__((always_inline))
COPER_HOST_INLINE int pack_ids_body(const int64_t* src, int64_t n, int32_t* dst, const int64_t* indptr, int64_t n_rows, int32_t* status) {
  int32_t st = 0;
  int64_t bad = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t v = src[i];
    const int32_t w = (int32_t)v;
    bad |= v ^ (int64_t)w;                     // (non-zero iff v is outside [-2^31, 2^31))
    dst[i] = w;
  }
  if (bad) st |= 1;
  if (n_rows > 0) {
    if (indptr[0] != 0 || indptr[n_rows] != n) return COPER_EINVAL;
    // every row ascending <=> every step down of the flat array sits on a row boundary: two flat loops (a loop per row spends
    // its time in the prologue of a five-element vector loop: 380 us per 100 K entries against 20)
    int64_t down = 0;
    for (int64_t i = 1; i < n; ++i) down += src[i] < src[i - 1];
    int64_t down_b = 0, malformed = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
      const int64_t lo = indptr[r], hi = indptr[r + 1];
      malformed |= (lo < 0) | (hi < lo) | (hi > n);
    }
    if (malformed) return COPER_EINVAL;
    if (n >= 2)
      for (int64_t r = 0; r < n_rows; ++r) {      // (branch-free: whether a row starts below its predecessor's end is a coin toss)
        const int64_t lo = indptr[r], hi = indptr[r + 1];
        const int64_t ok = (int64_t)(lo > 0) & (int64_t)(lo < n) & (int64_t)(hi > lo);      // a non-empty row's start: each boundary once
        const int64_t j = ok ? lo : 1;
        down_b += ok & (int64_t)(src[j] < src[j - 1]);
      }
    if (down != down_b) st |= 2;
  }
  *status = st;
  return COPER_OK;
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static int pack_ids_avx2(const int64_t* src, int64_t n, int32_t* dst, const int64_t* indptr, int64_t n_rows, int32_t* status) {
  return pack_ids_body(src, n, dst, indptr, n_rows, status);
}
#endif
static int pack_ids_generic(const int64_t* src, int64_t n, int32_t* dst, const int64_t* indptr, int64_t n_rows, int32_t* status) {
  return pack_ids_body(src, n, dst, indptr, n_rows, status);
}

COPER_API int coper_pack_ids_i32(const int64_t* src, int64_t n, int32_t* dst, const int64_t* indptr, int64_t n_rows, int32_t* status) {
  if (n < 0 || n_rows < 0 || !status || (n > 0 && (!src || !dst)) || (n_rows > 0 && !indptr)) return COPER_EINVAL;
#if defined(__x86_64__)
  if (__builtin_cpu_supports("avx2")) return pack_ids_avx2(src, n, dst, indptr, n_rows, status);
#endif
  return pack_ids_generic(src, n, dst, indptr, n_rows, status);
}

// ---- Hits@k / mean rank / MRR of a pass's ranks on the host (metrics.py:53-57,65-76) ----
// np.mean over float64 is numpy's pairwise sum (blocks of <= 128 elements, eight running sums, halves split at multiples of 8)
// applied to chunks of 8,192 elements (its reduction buffer, np.getbufsize()) whose sums are added up in order; the same order
// here, so the two means are the float64 values np.mean(ranks) and np.mean(1.0 / ranks) give (checked against NumPy in the tests).
COPER_HOST_INLINE double np_pairwise_sum(const double* a, int64_t n) {
  if (n < 8) {
    double r = 0.;
    for (int64_t i = 0; i < n; ++i) r += a[i];
    return r;
  }
  // (iterative over the 128-element leaves would change nothing: the recursion's depth is log2(8192 / 128) = 6)
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

COPER_HOST_INLINE int hits_means_body(const int32_t* ranks, int64_t n, const int32_t* levels, int32_t n_levels, double* mean_rank, double* mrr,
                                      double* hits, std::vector<double>& tab, std::vector<double>& inv) {
  // plain loops the compiler vectorises: extremes and the integer sum, one counter per level; then 1 / rank from a table of the
  // float64 quotients (grown on demand, kept per thread) -- a division per element was 50 of this function's 74 us
  int32_t lo = ranks[0], hi = ranks[0];
  int64_t sum = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t r = ranks[i];
    lo = r < lo ? r : lo;
    hi = r > hi ? r : hi;
    sum += r;
  }
  if (lo < 1) return COPER_EINVAL;              // (ranks start at 1: metrics.py:50)
  for (int32_t k = 0; k < n_levels; ++k) {
    const int32_t lv = levels[k];
    int32_t c = 0;                              // (n < 2^31)
    for (int64_t i = 0; i < n; ++i) c += ranks[i] <= lv;
    hits[k] = (double)c / (double)n;
  }
  // (the table pays when it is small beside the ranks it serves: at most 2^20 entries -- 8 MB per thread, the NumPy fallback's cap
  //  is 2^22 -- and at most four per rank; ADVICE r5: 4,096 ranks of a 10M-entity evaluation filled 80 MB to save 4,096 divisions)
  if ((int64_t)tab.size() <= (int64_t)hi && hi < (1 << 20) && (int64_t)hi <= 4 * n) {
    const size_t old_n = tab.size() < 1 ? 1 : tab.size();
    size_t want = (size_t)hi + 1 > 2 * old_n ? (size_t)hi + 1 : 2 * old_n;
    if (want > ((size_t)1 << 20)) want = (size_t)1 << 20;
    tab.resize(want);
    tab[0] = 0.;
    for (size_t v = old_n; v < tab.size(); ++v) tab[v] = 1.0 / (double)v;
  }
  inv.resize((size_t)n);
  double* const ip = inv.data();
  if ((int64_t)tab.size() > (int64_t)hi) {
    const double* t = tab.data();
    for (int64_t i = 0; i < n; ++i) ip[i] = t[ranks[i]];
  } else {
    for (int64_t i = 0; i < n; ++i) ip[i] = 1.0 / (double)ranks[i];
  }
  *mean_rank = (double)sum / (double)n;         // (integers below 2^53: exact in any order, the value np.mean has)
  double acc = 0.;
  for (int64_t i = 0; i < n; i += 8192) acc += np_pairwise_sum(ip + i, n - i < 8192 ? n - i : 8192);
  *mrr = acc / (double)n;
  return COPER_OK;
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static int hits_means_avx2(const int32_t* ranks, int64_t n, const int32_t* levels, int32_t n_levels, double* mean_rank,
                                                           double* mrr, double* hits, std::vector<double>& tab, std::vector<double>& inv) {
  return hits_means_body(ranks, n, levels, n_levels, mean_rank, mrr, hits, tab, inv);
}
#endif
static int hits_means_generic(const int32_t* ranks, int64_t n, const int32_t* levels, int32_t n_levels, double* mean_rank, double* mrr,
                              double* hits, std::vector<double>& tab, std::vector<double>& inv) {
  return hits_means_body(ranks, n, levels, n_levels, mean_rank, mrr, hits, tab, inv);
}

COPER_API int coper_hits_means(const int32_t* ranks, int64_t n, const int32_t* levels, int32_t n_levels, double* mean_rank, double* mrr,
                               double* hits) {
  if (n <= 0 || n >= 0x7fffffff || n_levels < 0 || !ranks || (n_levels > 0 && (!levels || !hits)) || !mean_rank || !mrr) return COPER_EINVAL;
  static thread_local std::vector<double> tab, inv;      // (their addresses taken once: a thread_local access inside a loop is a call)
#if defined(__x86_64__)
  if (__builtin_cpu_supports("avx2")) return hits_means_avx2(ranks, n, levels, n_levels, mean_rank, mrr, hits, tab, inv);
#endif
  return hits_means_generic(ranks, n, levels, n_levels, mean_rank, mrr, hits, tab, inv);
}

// The audit ACTS: host logic only -- the caller brings the two words it read with coper_band_audit or
// received through coper_band_audit_post.
COPER_API int coper_band_policy(coper_handle* h, float max_ratio, int64_t n_pairs, int32_t* action, float* kappa_now) {
  if (!h) return COPER_EINVAL;
  int act = COPER_BAND_KEEP;
  if (h->band_consts && n_pairs > 0 && max_ratio == max_ratio && max_ratio > 0.5f) {
    // widen so that the error just seen sits at or below a QUARTER of the new allowance: x2 for (0.5, 1), x 2^ceil(log2(4 ratio))
    // from 1 on (the pass is then re-ranked under the new band by the caller); never beyond 2^20 (kappa = 1 is every pair)
    float mult = 2.f;
    if (max_ratio >= 1.f) {
      act = COPER_BAND_RERANK;
      mult = 4.f;
      while (mult < 4.f * max_ratio && mult < 1048576.f) mult *= 2.f;
    } else {
      act = COPER_BAND_WIDENED;
    }
    float m = h->band_kappa_mult * mult;
    const float base = h->cfg.rank_band_kappa > 0.f ? h->cfg.rank_band_kappa : COPER_BAND_KAPPA_DEFAULT;
    if (base * m > 1.f) m = 1.f / base;
    if (m < h->band_kappa_mult) m = h->band_kappa_mult;
    h->band_kappa_mult = m;
    h->band_launches = 0;      // the next count launch carries the audit again: the new band is checked at once
  }
  if (action) *action = act;
  if (kappa_now) *kappa_now = h->band_consts ? band_kappa(h) : 0.f;
  return COPER_OK;
}

COPER_API int coper_stale_passes(coper_handle* h, int64_t* n_passes, void* stream) {
  if (!h || !n_passes) return COPER_EINVAL;
  *n_passes = h->stale_passes_host;
  h->stale_passes_host = 0;
  if (!h->group_done) return COPER_OK;
  hipStream_t s = (hipStream_t)stream;
  int32_t v = 0;
  COPER_HIP_TRY(h, hipMemcpyAsync(&v, h->group_done + 2, sizeof v, hipMemcpyDeviceToHost, s));
  COPER_HIP_TRY(h, hipMemsetAsync(h->group_done + 2, 0, sizeof v, s));
  COPER_HIP_TRY(h, hipStreamSynchronize(s));
  *n_passes += v;
  return COPER_OK;
}

COPER_API int coper_check_ids(coper_handle* h, int64_t* n_bad, void* stream) {
  if (!h || !n_bad) return COPER_EINVAL;
  *n_bad = 0;
  if (!h->rel_count) return COPER_OK;
  int32_t v = 0;
  COPER_HIP_TRY(h, hipMemcpyAsync(&v, h->rel_count + h->dm.R + 1, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream));
  COPER_HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
  *n_bad = v;
  return COPER_OK;
}

// ---- CRC-32C on the host (tf_bundle.py): hardware instruction when present, slicing-by-8 tables otherwise
static uint32_t g_crc_tab[8][256];
static void crc_tables_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
    g_crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFF];
}
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(uint32_t c, const unsigned char* p, uint64_t n) {
  while (n && ((uintptr_t)p & 7)) { c = __builtin_ia32_crc32qi(c, *p++); --n; }
  uint64_t c64 = c;
  for (; n >= 8; n -= 8, p += 8) { uint64_t v; memcpy(&v, p, 8); c64 = __builtin_ia32_crc32di(c64, v); }
  c = (uint32_t)c64;
  while (n--) c = __builtin_ia32_crc32qi(c, *p++);
  return c;
}
#endif
static uint32_t crc32c_sw(uint32_t c, const unsigned char* p, uint64_t n) {
  static std::once_flag once;
  std::call_once(once, crc_tables_init);
  for (; n >= 8; n -= 8, p += 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = g_crc_tab[7][lo & 0xFF] ^ g_crc_tab[6][(lo >> 8) & 0xFF] ^ g_crc_tab[5][(lo >> 16) & 0xFF] ^ g_crc_tab[4][lo >> 24] ^
        g_crc_tab[3][hi & 0xFF] ^ g_crc_tab[2][(hi >> 8) & 0xFF] ^ g_crc_tab[1][(hi >> 16) & 0xFF] ^ g_crc_tab[0][hi >> 24];
  }
  while (n--) c = g_crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c;
}

COPER_API uint32_t coper_crc32c(uint32_t crc, const void* data, uint64_t n) {
  uint32_t c = crc ^ 0xFFFFFFFFu;
  const unsigned char* p = (const unsigned char*)data;
  static const int force_sw = getenv("COPER_CRC_SOFTWARE") != nullptr;   // tests: the table path on a CPU that has the instruction
#if defined(__x86_64__)
  if (!force_sw && __builtin_cpu_supports("sse4.2")) return crc32c_hw(c, p, n) ^ 0xFFFFFFFFu;
#endif
  return crc32c_sw(c, p, n) ^ 0xFFFFFFFFu;
}

COPER_API int64_t coper_live_device_bytes(void) {
  std::lock_guard<std::mutex> lk(g_ledger_mu);
  return g_ledger_bytes;
}

COPER_API int coper_profile_enable(coper_handle* h, int enable) {
  if (!h) return COPER_EINVAL;
  h->profile = enable != 0;
  return COPER_OK;
}

COPER_API int coper_profile_read(coper_handle* h, const char* kernel, double* total_ms, int64_t* launches) {
  if (!h || !kernel) return COPER_EINVAL;
  Timer& t = h->timers[kernel];
  for (auto& p : t.pending) {
    COPER_HIP_TRY(h, hipEventSynchronize(p.second));
    float ms = 0.f;
    COPER_HIP_TRY(h, hipEventElapsedTime(&ms, p.first, p.second));
    t.total_ms += ms;
    t.launches += 1;
    h->event_pool.push_back(p.first);
    h->event_pool.push_back(p.second);
  }
  t.pending.clear();
  if (total_ms) *total_ms = t.total_ms;
  if (launches) *launches = t.launches;
  t.total_ms = 0.0;
  t.launches = 0;
  return COPER_OK;
}

}  // extern "C"
