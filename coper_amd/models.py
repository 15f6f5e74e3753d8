"""`ConvE` -- host-side mirror of `qa_cpg.models.ConvE` (CoPER_ConvE/qa_cpg/models.py:97-201) over
libcoper_hip.so.

Same constructor contract (`ConvE(model_descriptors)`, keys of models.py:99-130), same variable
names (`.variables[...]`, models.py:316-325), same fetch attributes the callers use
(`.e1 .e2 .rel .e2_multi .obj_lookup_values .predicted_e2_emb .predictions_all
.predictions_lookup .is_train .input_iterator_handle`, models.py:135-190), evaluated through a
`session.run(fetches, feed_dict)` shim (`ConvE.session()`), so `run_cpg.py:_evaluate` /
`metrics.ranking_and_hits` read the same.  Training: `train_init()` / `train_step(batch)`.

torch is plumbing (device memory, streams); all compute is in the HIP library.  There is no CPU
path: constructing a model without a GPU or without the built library raises."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib

__all__ = ["ConvE", "ContextualParameterGenerator", "ParameterLookup", "OutOfRangeError"]


class OutOfRangeError(Exception):
    """End of an evaluation pass (the reference sees `tf.errors.OutOfRangeError`, metrics.py:59)."""


class ContextualParameterGenerator(object):
    """Parameter holder mirroring models.py:32-54: `.projections` = list of [in, n] matrices named
    '<name>/CPG/Projection<i>'.  `generate` happens on the device inside `coper_prepare`."""

    def __init__(self, name, shape, projections, batch_norms):
        self.name, self.shape, self.projections, self.batch_norms = name, list(shape), projections, batch_norms


class ParameterLookup(object):
    """Parameter holder mirroring models.py:79-88: `.param_lookup_matrix` [num_rel, prod(shape)]."""

    def __init__(self, name, output_shape, matrix):
        self.name, self.output_shape, self.param_lookup_matrix = name, list(output_shape), matrix


class _Fetch(object):
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<coper fetch %s>" % self.name


class _TrainOp(_Fetch):
    """`model.train_op` (models.py:196-200): a fetch for `session.run((model.loss, model.train_op), {model.is_train: True,
    ...})` as `run_cpg.py:211-219` issues it, and callable on a batch for loops that hold the batches themselves."""

    def __init__(self, model):
        _Fetch.__init__(self, "train_op")
        self._model = model

    def __call__(self, batch):
        return self._model.train_step(batch)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _host(v):
    """A batch field as a host NumPy array (`session.run` returns host arrays); device tensors are copied back."""
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


class ConvE(object):
    def __init__(self, model_descriptors: dict, device=None, shard=None, score_mode="f32", rank_band_kappa=0.0, band_audit_period=0,
                 role="both", rel_mod=None):
        md = dict(model_descriptors)
        # required keys, as models.py:99-105,119-130 reads them
        for key in ("use_negative_sampling", "label_smoothing_epsilon", "num_ent", "num_rel", "ent_emb_size",
                    "rel_emb_size", "input_dropout", "hidden_dropout", "output_dropout", "add_loss_summaries",
                    "add_variable_summaries", "add_tensor_summaries", "learning_rate"):
            if key not in md:
                raise KeyError(key)
        self.model_descriptors = md
        self.use_negative_sampling = md["use_negative_sampling"]
        self.num_ent, self.num_rel = int(md["num_ent"]), int(md["num_rel"])
        self.ent_emb_size, self.rel_emb_size = int(md["ent_emb_size"]), int(md["rel_emb_size"])
        self.is_parameter_lookup = md.get("do_parameter_lookup", False)
        self.context_rel_conv = md.get("context_rel_conv", None)
        self.context_rel_out = md.get("context_rel_out", None)
        self.concat_rel = md.get("concat_rel", False)

        if not torch.cuda.is_available():
            raise RuntimeError("coper_amd.ConvE needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.shard = (0, self.num_ent) if shard is None else (int(shard[0]), int(shard[1]))
        self._lib = _lib.load()
        if score_mode not in ("f32", "bf16x3"):
            raise ValueError("score_mode: 'f32' (exact-f32 MFMA) or 'bf16x3' (split-bf16 operands, 3 bf16 MFMAs per product)")
        mode = {"f32": _lib.SCORE_F32, "bf16x3": _lib.SCORE_BF16X3}[score_mode]
        self.score_mode = score_mode
        if role not in ("both", "encode", "score"):
            raise ValueError("role: 'both', 'encode' (no entity planes; encode only) or 'score' (no generated weights; scoring only)")
        self.role = role
        cfg = _lib.make_config(md, device=self.device.index or 0, shard=self.shard, score_mode=mode, rank_band_kappa=rank_band_kappa,
                               band_audit_period=band_audit_period, role={"both": _lib.ROLE_BOTH, "encode": _lib.ROLE_ENCODE, "score": _lib.ROLE_SCORE}[role],
                               rel_mod=rel_mod)
        self.rel_mod = (int(rel_mod[0]), int(rel_mod[1])) if rel_mod is not None else None
        h = C.c_void_p()
        rc = self._lib.coper_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise _lib.CoperError(rc, (self._lib.coper_last_error(None) or b"").decode())
        self._h = h
        F, Ho, Wo, nl = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int64()
        self._lib.coper_get_dims(h, C.byref(F), C.byref(Ho), C.byref(Wo), C.byref(nl))
        self.fc_input_size, self.conv_out_height, self.conv_out_width, self.n_local = F.value, Ho.value, Wo.value, nl.value

        # parameter specs from the library (single source of truth for the shapes)
        self._specs = {}
        for i in range(self._lib.coper_num_params(h)):
            name, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
            self._lib.coper_param_spec(h, i, C.byref(name), shape, C.byref(nd))
            self._specs[name.value.decode()] = tuple(shape[j] for j in range(nd.value))
        self._tensors: Dict[str, torch.Tensor] = {}
        self._prepared = False

        # fetch handles (models.py:135-190)
        self.input_iterator_handle = _Fetch("input_iterator_handle")
        self.is_train = _Fetch("is_train")
        self.e1, self.e2, self.rel = _Fetch("e1"), _Fetch("e2"), _Fetch("rel")
        self.e2_multi = _Fetch("e2_multi")
        self.obj_lookup_values = _Fetch("lookup_values") if self.use_negative_sampling else None
        self.predicted_e2_emb = _Fetch("predicted_e2_emb")
        self.predictions_all = _Fetch("predictions_all")
        self.predictions_lookup = _Fetch("predictions_lookup")
        self.loss = _Fetch("loss")                 # models.py:192
        self.train_op = _TrainOp(self)             # models.py:196-200
        self.summaries = None

    # ---------------------------------------------------------------- parameters
    @property
    def parameter_specs(self) -> Dict[str, tuple]:
        """leaf name -> LOCAL shape (ent_emb / pred_bias rows are the shard's)."""
        return dict(self._specs)

    def load_parameters(self, params: Dict[str, "np.ndarray | torch.Tensor"], global_rows=True):
        """Sets parameters by the reference's leaf names.  `global_rows`: ent_emb / pred_bias are given
        for all |E| entities and sliced to the shard here."""
        lo, hi = self.shard
        for name, want in self._specs.items():
            if name not in params:
                continue
            v = params[name]
            t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))
            if name in ("ent_emb", "pred_bias") and global_rows and t.shape[0] == self.num_ent and (lo, hi) != (0, self.num_ent):
                if name == "ent_emb" and self.score_mode == "bf16x3":
                    # every shard of one table scales its planes by the same power of two (include/coper_hip.h: x3_ent_absmax)
                    self.set_x3_ent_absmax(float(t.abs().max()) if t.numel() else 0.0)
                t = t[lo:hi]
            elif name == "ent_emb" and self.score_mode == "bf16x3" and getattr(self, "_x3_absmax", None) is not None:
                # new rows for this shard alone: a table-wide maximum agreed for the OLD rows may no longer cover them (coper_prepare
                # would refuse on this rank only while the others go on into collectives: ADVICE r4).  Back to the shard's own
                # maximum; an EntityShardedRanker re-agrees on the table-wide one with its next chunk (sharding.py step 1)
                _lib.check(self._h, self._lib.coper_set_x3_ent_absmax(self._h, 0.0))
                self._x3_absmax = None
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            if t.numel() != int(np.prod(want)):
                raise ValueError("parameter %s: got shape %s, need %s" % (name, tuple(t.shape), want))
            self._tensors[name] = t
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            _lib.check(self._h, self._lib.coper_set_param(self._h, name.encode(), _ptr(t), shape, t.dim()))
        if "ent_emb" in params:
            self._ent_absmax_cache = None
        self._prepared = False
        return self

    @property
    def variables(self):
        """Dict with the keys of models.py:316-325; generated parameters are holder objects like the
        reference's generator objects."""
        T = self._tensors
        out = {"ent_emb": T.get("ent_emb"), "pred_bias": T.get("pred_bias")}
        if not self.is_parameter_lookup:
            out["rel_emb"] = T.get("rel_emb")
        C_ = int(self.model_descriptors.get("conv_num_channels", 32))
        fh = int(self.model_descriptors.get("conv_filter_height", 3))
        fw = int(self.model_descriptors.get("conv_filter_width", 3))
        shapes = {"conv1_weights": [fh, fw, 1, C_], "conv1_bias": [C_],
                  "fc_weights": [self.fc_input_size, self.ent_emb_size], "fc_bias": [self.ent_emb_size]}
        for name, ctx in (("conv1_weights", self.context_rel_conv), ("conv1_bias", self.context_rel_conv),
                          ("fc_weights", self.context_rel_out), ("fc_bias", self.context_rel_out)):
            if ctx is None:
                out[name] = T.get(name)
            elif self.is_parameter_lookup:
                out[name] = ParameterLookup(name, shapes[name], T.get(name))
            else:
                proj = [T.get("%s/CPG/Projection%d" % (name, i)) for i in range(len(ctx) + 1)]
                bns = [{leaf: T.get("%s/CPG/Projection%d/BatchNorm/%s" % (name, i, leaf))
                        for leaf in ("gamma", "beta", "moving_mean", "moving_variance")} for i in range(len(ctx))]
                out[name] = ContextualParameterGenerator(name, shapes[name], proj, bns)
        return out

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def prepare(self):
        """coper_prepare: fold BN, evaluate the generators once per relation, build the MFMA images."""
        missing = [n for n in self._specs if n not in self._tensors]
        if missing:
            raise _lib.CoperError(2, "parameters never set: %s" % ", ".join(missing))
        with torch.cuda.device(self.device):
            _lib.check(self._h, self._lib.coper_prepare(self._h, self._stream()))
        self._prepared = True
        return self

    def reserve(self, max_queries, max_filter_nnz=0):
        with torch.cuda.device(self.device):
            _lib.check(self._h, self._lib.coper_reserve(self._h, int(max_queries), int(max_filter_nnz), self._stream()))
        return self

    # ---------------------------------------------------------------- compute
    def _ids(self, a, dtype=torch.int64):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device, non_blocking=True)

    def _need_prepared(self):
        if not self._prepared:
            self.prepare()

    def widen_ids(self, src: torch.Tensor, out: Optional[torch.Tensor] = None):
        """int32 ids -> the int64 the C-ABI takes, on the device (coper_widen_ids).  `src`: an int32 tensor on the model's
        device, or a PINNED host tensor -- the kernel then reads it over PCIe itself (one launch instead of a copy-engine
        transfer and a widening pass)."""
        if src.dtype != torch.int32 or not src.is_contiguous():
            raise ValueError("widen_ids: a contiguous int32 tensor")
        if src.device.type == "cpu" and not src.is_pinned():
            raise ValueError("widen_ids: host tensors must be pinned (device-mapped) memory")
        n = src.numel()
        if out is None:
            out = torch.empty(n, dtype=torch.int64, device=self.device)
        elif out.dtype != torch.int64 or out.numel() != n or not out.is_contiguous() or out.device != self.device:
            raise ValueError("widen_ids: out must be a contiguous int64 tensor of %d elements on %s" % (n, self.device))
        _lib.check(self._h, self._lib.coper_widen_ids(self._h, C.c_void_p(src.data_ptr()), n, _ptr(out), self._stream()))
        return out

    def stage_next(self, src: torch.Tensor, out: torch.Tensor):
        """coper_stage_ids_next: the pinned int32 batch `src` is brought in and widened into `out` (int64, device) BESIDE the
        encoder launch of the next encode / rank_pass call on this model -- the staging of pass n + 1 under pass n's kernels.
        `out` must not be a buffer that call reads."""
        if src.dtype != torch.int32 or not src.is_contiguous() or (src.device.type == "cpu" and not src.is_pinned()):
            raise ValueError("stage_next: a contiguous int32 tensor, pinned when on the host")
        if out.dtype != torch.int64 or out.numel() != src.numel() or not out.is_contiguous() or out.device != self.device:
            raise ValueError("stage_next: out must be a contiguous int64 tensor of %d elements on %s" % (src.numel(), self.device))
        self._stage_keep = (src, out)      # (the job is carried out later: keep both alive)
        _lib.check(self._h, self._lib.coper_stage_ids_next(self._h, C.c_void_p(src.data_ptr()), src.numel(), _ptr(out)))

    def group_next(self, e1: Optional[torch.Tensor], rel: torch.Tensor, e1_rows: bool = False):
        """coper_group_next: `e1` / `rel` (int64, device -- typically views of the staging array a `stage_next` job fills) are the id
        arrays of the pass AFTER the next one; that batch is sorted by relation by one more workgroup of the next pass's encoder
        launch, and the pass that then comes with exactly these tensors starts with its encoder.  Results never depend on it."""
        if rel is None or rel.dtype != torch.int64 or not rel.is_contiguous() or rel.device != self.device:
            raise ValueError("group_next: rel must be a contiguous int64 tensor on %s" % self.device)
        if not e1_rows and (e1 is None or e1.dtype != torch.int64 or not e1.is_contiguous() or e1.device != self.device or e1.numel() != rel.numel()):
            raise ValueError("group_next: e1 must be a contiguous int64 tensor of rel's size on %s" % self.device)
        self._group_keep = (e1, rel)
        _lib.check(self._h, self._lib.coper_group_next(self._h, None if e1_rows else _ptr(e1), _ptr(rel), rel.numel(), 1 if e1_rows else 0))

    def stage_batch(self, *arrays):
        """Host (NumPy) id arrays of one batch -> int64 tensors on the device through ONE pinned int32 buffer and one launch of
        coper_widen_ids (instead of one pageable H2D copy per array).  Arrays whose ids do not fit int32, and tensors, take
        the ordinary route.  Two pinned buffers alternate; a buffer is reused when the launch that read it has finished."""
        if not arrays or any(isinstance(a, torch.Tensor) for a in arrays):
            return tuple(self._ids(a) for a in arrays)
        arrs = [np.ascontiguousarray(a).reshape(-1) for a in arrays]
        if any(a.dtype.kind not in "iu" or (a.size and (int(a.max()) >= 2 ** 31 or int(a.min()) < -2 ** 31)) for a in arrs):
            return tuple(self._ids(a) for a in arrays)
        total = sum(a.size for a in arrs)
        if total == 0:
            return tuple(self._ids(a) for a in arrays)
        st = getattr(self, "_stage", None)
        if st is None:
            st = self._stage = {"pin": [None, None], "ev": [torch.cuda.Event(), torch.cuda.Event()], "turn": 0}
        k = st["turn"]
        st["turn"] = 1 - k
        if st["pin"][k] is None or st["pin"][k].numel() < total:
            st["ev"][k].synchronize()
            st["pin"][k] = torch.empty(max(total, 1 << 16), dtype=torch.int32).pin_memory()
        else:
            st["ev"][k].synchronize()          # the launch that read this buffer two calls ago
        pin = st["pin"][k]
        host = pin.numpy()
        offs, o = [], 0
        for a in arrs:
            host[o:o + a.size] = a               # (NumPy converts int64 -> int32 while copying)
            offs.append(o)
            o += a.size
        dev = torch.empty(total, dtype=torch.int64, device=self.device)
        self.widen_ids(pin[:total], out=dev)
        st["ev"][k].record(torch.cuda.current_stream(self.device))
        return tuple(dev[o:o + a.size] for o, a in zip(offs, arrs))

    def stage_csr(self, e2, filt_indptr, filt_idx):
        """The targets and the CSR filter of a host batch (int64 NumPy arrays) -> int64 device tensors, through the pinned int32 buffer
        of `stage_batch` -- with the narrowing copy, the range check and the rows-ascending check of the filter done in ONE native
        pass per array (coper_pack_ids_i32) instead of NumPy passes.  None when an id does not fit int32 or a filter row is not
        ascending: the caller takes the general route (`canonical_csr`, `stage_batch`)."""
        arrs = (e2, filt_indptr, filt_idx)
        if any(not isinstance(a, np.ndarray) or a.dtype != np.int64 or a.ndim != 1 or not a.flags.c_contiguous for a in arrs):
            return None
        B, nnz = e2.size, filt_idx.size
        if filt_indptr.size != B + 1 or B == 0:
            return None
        total = B + B + 1 + nnz
        st = getattr(self, "_stage", None)
        if st is None:
            st = self._stage = {"pin": [None, None], "ev": [torch.cuda.Event(), torch.cuda.Event()], "turn": 0}
        k = st["turn"]
        st["turn"] = 1 - k
        st["ev"][k].synchronize()                # the launch that read this buffer two calls ago
        if st["pin"][k] is None or st["pin"][k].numel() < total:
            st["pin"][k] = torch.empty(max(total, 1 << 16), dtype=torch.int32).pin_memory()
        pin = st["pin"][k]
        base = pin.data_ptr()
        status = C.c_int32()
        bad = 0
        for a, off, ip, rows in ((e2, 0, None, 0), (filt_indptr, B, None, 0), (filt_idx, 2 * B + 1, filt_indptr, B)):
            if self._lib.coper_pack_ids_i32(C.c_void_p(a.ctypes.data), a.size, C.c_void_p(base + 4 * off),
                                            C.c_void_p(ip.ctypes.data) if ip is not None else None, rows, C.byref(status)) != 0:
                return None
            bad |= status.value
        if bad:
            return None
        dev = torch.empty(total, dtype=torch.int64, device=self.device)
        self.widen_ids(pin[:total], out=dev)
        st["ev"][k].record(torch.cuda.current_stream(self.device))
        return dev[:B], dev[B:2 * B + 1], dev[2 * B + 1:total]

    def stage_persistent(self, e1, rel, e2, filt_indptr, filt_idx):
        """A host batch marshalled ONCE for repeated evaluation (an `EvalDataset` scored after every epoch, run_cpg.py:18-35):
        [e1 | rel | e2 | filt_indptr | filt_idx] as int32 in one pinned buffer of its own, the int64 device arrays the C-ABI
        takes, and pinned memory for the ranks.  `rank_pass_staged` brings the batch in with one launch per pass and posts
        the ranks back; the host does no per-pass marshalling.  None when an id does not fit int32."""
        arrs = [np.ascontiguousarray(np.asarray(a)).reshape(-1) for a in (e1, rel, e2, filt_indptr, filt_idx)]
        if any(a.dtype.kind not in "iu" or (a.size and (int(a.max()) >= 2 ** 31 or int(a.min()) < -2 ** 31)) for a in arrs):
            return None
        sizes = [int(a.size) for a in arrs]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        total = max(1, int(offs[-1]))
        pin = torch.empty(total, dtype=torch.int32).pin_memory()
        host = pin.numpy()
        for a, o in zip(arrs, offs):
            host[o:o + a.size] = a
        B = sizes[0]
        sb = dict(pin=pin, total=int(offs[-1]), B=B, nnz=sizes[4],
                  stage=torch.empty(total, dtype=torch.int64, device=self.device),
                  ranks=torch.empty(max(1, B), dtype=torch.int32, device=self.device),
                  out_host=torch.empty(max(1, B) + 2, dtype=torch.int32).pin_memory(),      # (+ 2: the band audit's words)
                  event=torch.cuda.Event())
        sb["views"] = tuple(sb["stage"][o:o + n] for o, n in zip(offs, sizes))
        return sb

    def rank_pass_staged(self, sb):
        """One evaluation pass over a batch from `stage_persistent`: ids + CSR in over PCIe (coper_widen_ids), the fused pass
        (coper_encode_rank, ranks only), the int32 ranks out to pinned memory (coper_copy_out_i32) -- all asynchronous.
        Returns the pinned int32 tensor of the ranks, valid after `sb["event"].synchronize()`."""
        self._need_prepared()
        B = sb["B"]
        if B == 0:
            return sb["out_host"][:0]
        # the int64 device arrays are a copy of the pinned buffer, which is a copy of the caller's arrays taken when the dataset was
        # staged: nothing can have changed in either since the last pass, so only the FIRST pass brings the batch in (round 5:
        # -20 us per call of ranking_and_hits on a dataset that is scored after every epoch, run_cpg.py:228-250)
        if not sb.get("resident"):
            self.widen_ids(sb["pin"][:sb["total"]], out=sb["stage"][:sb["total"]])
            sb["resident"] = True
        e1, rel, e2, ip, ix = sb["views"]
        ranks = sb["ranks"][:B]
        _lib.check(self._h, self._lib.coper_encode_rank(self._h, _ptr(e1), _ptr(rel), None, _ptr(e2), _ptr(ip), _ptr(ix),
                                                        sb["nnz"], B, None, _ptr(ranks), None, self._stream()))
        # the ranks and, behind them, the audit's two words (read and reset) in one launch: coper_post_ranks_audit
        _lib.check(self._h, self._lib.coper_post_ranks_audit(self._h, _ptr(ranks), B, C.c_void_p(sb["out_host"].data_ptr()), 1, self._stream()))
        sb["event"].record(torch.cuda.current_stream(self.device))
        return sb["out_host"][:B]

    def copy_out(self, src: torch.Tensor, dst: torch.Tensor):
        """int32 results (the ranks of a pass) from the device to `dst` -- a pinned host tensor (or a device tensor): one small
        launch right behind the pass's last kernel (coper_copy_out_i32).  Synchronise with the stream before reading dst."""
        if src.dtype != torch.int32 or dst.dtype != torch.int32 or src.numel() != dst.numel() or not (src.is_contiguous() and dst.is_contiguous()):
            raise ValueError("copy_out: two contiguous int32 tensors of one size")
        if src.device != self.device or (dst.device.type == "cpu" and not dst.is_pinned()):
            raise ValueError("copy_out: src on %s, dst there or in pinned host memory" % self.device)
        _lib.check(self._h, self._lib.coper_copy_out_i32(self._h, _ptr(src), src.numel(), C.c_void_p(dst.data_ptr()), self._stream()))
        return dst

    def post_next(self, src: torch.Tensor, dst: torch.Tensor):
        """coper_post_i32_next: `src` (int32, device: the ranks of the pass just queued) is copied to `dst` (pinned host memory, or
        the device) beside the first launch of the NEXT encode / rank_pass on this model instead of by a launch of its own.  Follow
        the last pass of a loop with `copy_out(src, dst)`; synchronise before reading dst."""
        if src.dtype != torch.int32 or dst.dtype != torch.int32 or src.numel() != dst.numel() or not (src.is_contiguous() and dst.is_contiguous()):
            raise ValueError("post_next: two contiguous int32 tensors of one size")
        if src.device != self.device or (dst.device.type == "cpu" and not dst.is_pinned()):
            raise ValueError("post_next: src on %s, dst there or in pinned host memory" % self.device)
        self._post_keep = (src, dst)
        _lib.check(self._h, self._lib.coper_post_i32_next(self._h, _ptr(src), src.numel(), C.c_void_p(dst.data_ptr())))

    def post_flush(self):
        """The end of a loop of `post_next` passes: the registration of the last one is withdrawn and its copy issued now
        (coper_copy_out_i32).  Synchronise before reading the destination."""
        keep = getattr(self, "_post_keep", None)
        if keep is None:
            return
        self._post_keep = None
        _lib.check(self._h, self._lib.coper_post_i32_next(self._h, None, 0, None))
        self.copy_out(*keep)

    def gather_entities(self, ids):
        self._need_prepared()
        ids = self._ids(ids)
        out = torch.empty((ids.numel(), self.ent_emb_size), device=self.device, dtype=torch.float32)
        _lib.check(self._h, self._lib.coper_gather_entities(self._h, _ptr(ids), ids.numel(), _ptr(out), self._stream()))
        return out

    def encode(self, e1, rel, e1_rows: Optional[torch.Tensor] = None):
        """predicted_e2_emb [B, d] (models.py:183)."""
        self._need_prepared()
        if e1 is not None and not isinstance(e1, torch.Tensor) and not isinstance(rel, torch.Tensor):
            e1, rel = self.stage_batch(e1, rel)          # host ids: one pinned int32 buffer, one launch (not two pageable copies)
        rel = self._ids(rel)
        e1 = self._ids(e1) if e1 is not None else None
        B = rel.numel()
        out = torch.empty((B, self.ent_emb_size), device=self.device, dtype=torch.float32)
        if e1_rows is not None:
            e1_rows = e1_rows.to(device=self.device, dtype=torch.float32).contiguous()
        _lib.check(self._h, self._lib.coper_encode(self._h, _ptr(e1), _ptr(rel), B, _ptr(e1_rows), _ptr(out), self._stream()))
        return out

    def score_all(self, h):
        """predictions_all [B, n_local] logits (models.py:434-437)."""
        self._need_prepared()
        B = h.shape[0]
        out = torch.empty((B, self.n_local), device=self.device, dtype=torch.float32)
        _lib.check(self._h, self._lib.coper_score_all(self._h, _ptr(h), B, _ptr(out), self.n_local, self._stream()))
        return out

    def score_lookup(self, h, lookup):
        """predictions_lookup [B, L] (models.py:439-443)."""
        self._need_prepared()
        lookup = self._ids(lookup, torch.int32)
        B, L = lookup.shape
        out = torch.zeros((B, L), device=self.device, dtype=torch.float32)
        _lib.check(self._h, self._lib.coper_score_lookup(self._h, _ptr(h), _ptr(lookup), B, L, _ptr(out), self._stream()))
        return out

    def target_scores(self, h, e2):
        """[2, B]: row 0 the mode's logit of (b, e2[b]), bit-identical to what score_all writes for that element; row 1
        the fp32-chain logit of the same pair (what the bf16x3 mode's exact band decides close comparisons against; equal
        to row 0 in the f32 mode).  Entries whose e2 is not on this shard are 0: the sum over shards is the full result."""
        self._need_prepared()
        e2 = self._ids(e2)
        out = torch.empty((2, e2.numel()), device=self.device, dtype=torch.float32)
        _lib.check(self._h, self._lib.coper_target_scores(self._h, _ptr(h), _ptr(e2), e2.numel(), _ptr(out), self._stream()))
        return out

    def gather_bias(self, ids):
        """pred_bias[ids] restricted to the shard (0 for ids it does not hold): with gather_entities, what an entity-sharded
        evaluation all-reduces for the targets (sharding.py step 1).  Plain indexing of the registered tensor."""
        ids = self._ids(ids)
        lo, hi = self.shard
        own = (ids >= lo) & (ids < hi)
        b = self._tensors["pred_bias"]
        return torch.where(own, b[(ids - lo).clamp(0, hi - lo - 1)], torch.zeros((), device=self.device, dtype=torch.float32))

    def score_rows(self, h, rows, bias):
        """[B]: pred_bias_b + rows[b] . h[b] by the fp32 chain (coper_score_rows): the target logits from entity rows the
        caller holds."""
        self._need_prepared()
        rows = rows.to(device=self.device, dtype=torch.float32).contiguous()
        bias = bias.to(device=self.device, dtype=torch.float32).contiguous()
        B = rows.shape[0]
        out = torch.empty((B,), device=self.device, dtype=torch.float32)
        _lib.check(self._h, self._lib.coper_score_rows(self._h, _ptr(h), _ptr(rows), _ptr(bias), B, _ptr(out), self._stream()))
        return out

    def rank_counts(self, h, tgt, e2, filt_indptr, filt_idx, filt_nnz=None, k=0):
        """(n_greater, n_equal) int32 [B] over this shard (metrics.py:44-50 without logits); with k > 0 also
        the shard's top-k of the filtered row: (..., topk_val f32 [B,k], topk_idx int64 [B,k] global ids).
        `tgt`: the [2, B] tensor of target_scores (summed over shards)."""
        self._need_prepared()
        if tuple(tgt.shape) != (2, self._ids(e2).numel()):
            raise ValueError("rank_counts: tgt must be the [2, B] tensor target_scores returns")
        tgt = tgt.contiguous()
        e2, ip, ix = self._ids(e2), self._ids(filt_indptr), self._ids(filt_idx)
        B = e2.numel()
        nnz = int(ix.numel()) if filt_nnz is None else int(filt_nnz)
        ng = torch.empty((B,), device=self.device, dtype=torch.int32)
        ne = torch.empty((B,), device=self.device, dtype=torch.int32)
        tv = torch.empty((B, k), device=self.device, dtype=torch.float32) if k > 0 else None
        ti = torch.empty((B, k), device=self.device, dtype=torch.int64) if k > 0 else None
        _lib.check(self._h, self._lib.coper_rank_counts(self._h, _ptr(h), _ptr(tgt), _ptr(e2), _ptr(ip), _ptr(ix), nnz, B, int(k),
                                                        _ptr(ng), _ptr(ne), _ptr(tv), _ptr(ti), self._stream()))
        if k > 0:
            return ng, ne, tv, ti
        return ng, ne

    def pack_shard_record(self, ng, ne, tv=None, ti=None, reset_audit=True):
        """coper_pack_shard_record: the int64 record [B + 1, 1 + 2 k] this shard contributes to the exchange's all-gather (counts, top-k,
        and -- last row -- its band-audit words, read and reset on the device)."""
        B = int(ng.numel())
        k = int(tv.shape[1]) if tv is not None else 0
        rec = torch.empty((B + 1, 1 + 2 * k), dtype=torch.int64, device=self.device)
        _lib.check(self._h, self._lib.coper_pack_shard_record(self._h, _ptr(ng), _ptr(ne), _ptr(tv.contiguous()) if k else None,
                                                              _ptr(ti.contiguous()) if k else None, B, k, 1 if reset_audit else 0, _ptr(rec), self._stream()))
        return rec

    def merge_shard_records(self, allrec, world, B, k):
        """coper_merge_shard_records: (ranks int32 [B], n_equal int32 [B], cand_val f32 [B, world k] or None, cand_idx int64 or None)
        from the gathered records [world, B + 1, 1 + 2 k]."""
        allrec = allrec.contiguous()
        ranks = torch.empty((B,), dtype=torch.int32, device=self.device)
        ne = torch.empty((B,), dtype=torch.int32, device=self.device)
        vals = torch.empty((B, world * k), dtype=torch.float32, device=self.device) if k else None
        ids = torch.empty((B, world * k), dtype=torch.int64, device=self.device) if k else None
        _lib.check(self._h, self._lib.coper_merge_shard_records(self._h, _ptr(allrec), int(world), int(B), int(k), _ptr(ranks), _ptr(ne),
                                                                _ptr(vals), _ptr(ids), self._stream()))
        return ranks, ne, vals, ids

    def rank(self, h, e2, filt_indptr, filt_idx, filt_nnz=None, want_equal=True):
        """Filtered ranks int32 [B] (+ n_equal, or None with want_equal=False) on an unsharded model."""
        self._need_prepared()
        if not any(isinstance(a, torch.Tensor) for a in (e2, filt_indptr, filt_idx)):
            e2, filt_indptr, filt_idx = self.stage_batch(e2, filt_indptr, filt_idx)     # host batch: pinned int32 + one widening launch
        e2, ip, ix = self._ids(e2), self._ids(filt_indptr), self._ids(filt_idx)
        B = e2.numel()
        nnz = int(ix.numel()) if filt_nnz is None else int(filt_nnz)
        ranks = torch.empty((B,), device=self.device, dtype=torch.int32)
        ne = torch.empty((B,), device=self.device, dtype=torch.int32) if want_equal else None
        _lib.check(self._h, self._lib.coper_rank(self._h, _ptr(h), _ptr(e2), _ptr(ip), _ptr(ix), nnz, B, _ptr(ranks), _ptr(ne),
                                                 self._stream()))
        return ranks, ne

    def rank_pass(self, e1, rel, e2, filt_indptr, filt_idx, filt_nnz=None, want_equal=True, want_h=False, e1_rows=None, out=None):
        """One evaluation batch end to end (coper_encode_rank): what one `session.run` of the reference's ranker
        loop computes (metrics.py:40-57).  Returns (ranks int32 [B], n_equal or None[, h [B, d] when want_h]); same
        bits as encode() + rank().  In the bf16x3 mode without want_h the embedding never exists in fp32.  out: an int32 [B]
        device tensor to receive the ranks (pipelines that copy them out on another stream own their buffers)."""
        self._need_prepared()
        if e1 is not None and not any(isinstance(a, torch.Tensor) for a in (e1, rel, e2, filt_indptr, filt_idx)):
            e1, rel, e2, ip, ix = self.stage_batch(e1, rel, e2, filt_indptr, filt_idx)    # host batch: one pinned buffer, one launch
        else:
            rel = self._ids(rel)
            e1 = self._ids(e1) if e1 is not None else None
            e2, ip, ix = self._ids(e2), self._ids(filt_indptr), self._ids(filt_idx)
        B = rel.numel()
        nnz = int(ix.numel()) if filt_nnz is None else int(filt_nnz)
        if e1_rows is not None:
            e1_rows = e1_rows.to(device=self.device, dtype=torch.float32).contiguous()
        if out is not None and (out.dtype != torch.int32 or out.numel() != B or not out.is_contiguous() or out.device != self.device):
            raise ValueError("out: a contiguous int32 tensor of %d ranks on %s" % (B, self.device))
        ranks = out if out is not None else torch.empty((B,), device=self.device, dtype=torch.int32)
        ne = torch.empty((B,), device=self.device, dtype=torch.int32) if want_equal else None
        h = torch.empty((B, self.ent_emb_size), device=self.device, dtype=torch.float32) if want_h else None
        _lib.check(self._h, self._lib.coper_encode_rank(self._h, _ptr(e1), _ptr(rel), _ptr(e1_rows), _ptr(e2), _ptr(ip), _ptr(ix),
                                                        nnz, B, _ptr(h), _ptr(ranks), _ptr(ne), self._stream()))
        return (ranks, ne, h) if want_h else (ranks, ne)

    def capture_rank_pass(self, B, max_nnz, want_equal=True):
        """hipGraph capture of one encode -> fused-rank pass for batches of exactly B queries (the reference's
        per-`session.run` batch, B = 512, is launch-bound: ~14 kernel launches per batch).  Returns
        `run(e1, rel, e2, filt_indptr, filt_idx) -> (ranks, n_equal)`; the id / CSR arguments are copied into
        static device buffers (CSR padded to `max_nnz` entries), the graph is replayed, and the outputs are
        static tensors valid until the next replay.  Every coper_* call inside is allocation-free after
        `reserve`, which is what makes the sequence capturable.  want_equal=False: ranks only (what the reference
        computes) -- in the bf16x3 mode the shorter sequence with the fused tail kernel; n_equal is then None."""
        self._need_prepared()
        self.reserve(B, max_nnz)
        dev = self.device
        st = dict(e1=torch.zeros(B, dtype=torch.int64, device=dev), rel=torch.zeros(B, dtype=torch.int64, device=dev),
                  e2=torch.zeros(B, dtype=torch.int64, device=dev), ip=torch.zeros(B + 1, dtype=torch.int64, device=dev),
                  ix=torch.zeros(max(1, max_nnz), dtype=torch.int64, device=dev),
                  h=torch.empty((B, self.ent_emb_size), dtype=torch.float32, device=dev),
                  ranks=torch.empty(B, dtype=torch.int32, device=dev), ne=torch.empty(B, dtype=torch.int32, device=dev))

        def body():
            # one call per batch (the dense finalize writes h straight into the rank kernels' planes); the filter launch
            # is sized by max_nnz: entries past indptr[B] belong to no query and are skipped
            _lib.check(self._h, self._lib.coper_encode_rank(self._h, _ptr(st["e1"]), _ptr(st["rel"]), None, _ptr(st["e2"]),
                                                            _ptr(st["ip"]), _ptr(st["ix"]), max_nnz, B, _ptr(st["h"]),
                                                            _ptr(st["ranks"]), _ptr(st["ne"]) if want_equal else None,
                                                            self._stream()))

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                body()                      # warm-up outside capture (lazy attribute / workspace set-up)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()

        def run(e1, rel, e2, filt_indptr, filt_idx):
            ix = self._ids(filt_idx)
            n = int(ix.numel())
            if n > max_nnz:
                raise ValueError("filter has %d entries, graph was captured for %d" % (n, max_nnz))
            st["e1"].copy_(self._ids(e1)); st["rel"].copy_(self._ids(rel)); st["e2"].copy_(self._ids(e2))
            st["ip"].copy_(self._ids(filt_indptr))
            st["ix"][:n].copy_(ix)
            graph.replay()
            return st["ranks"], (st["ne"] if want_equal else None)

        run.graph = graph
        return run

    def ent_absmax(self) -> float:
        """The largest |ent_emb| element of the rows this handle holds (one pass over the shard, kept until the rows change)."""
        c = getattr(self, "_ent_absmax_cache", None)
        if c is None:
            t = self._tensors.get("ent_emb")
            c = self._ent_absmax_cache = float(t.abs().max()) if t is not None and t.numel() else 0.0
        return c

    def owned_rows(self, ids):
        """(ent_emb rows [n, d], pred_bias [n]) of GLOBAL ids this shard holds (sharding.py step 1): plain indexing of the registered
        tensors -- no prepared state needed, so a shard whose weights were just reloaded can answer before it is prepared again."""
        lo, hi = self.shard
        ids = self._ids(ids)
        loc = (ids - lo).clamp(0, max(0, hi - lo - 1))
        return self._tensors["ent_emb"].index_select(0, loc), self._tensors["pred_bias"].index_select(0, loc)

    def pack_owned_rows(self, local_rows: torch.Tensor, cap: int, hdr0: float, hdr1: float):
        """coper_pack_owned_rows: the [cap + 1, d + 1] buffer this shard contributes to step 1's all-gather (header row, then the
        rows / biases at `local_rows` -- int64 device tensor of shard-local row numbers --, zeros up to cap): one launch."""
        buf = torch.empty((cap + 1, self.ent_emb_size + 1), dtype=torch.float32, device=self.device)
        n = int(local_rows.numel())
        _lib.check(self._h, self._lib.coper_pack_owned_rows(self._h, _ptr(local_rows) if n else None, n, int(cap), float(hdr0), float(hdr1),
                                                            _ptr(buf), self._stream()))
        return buf

    def unpack_rows(self, gathered: torch.Tensor, take1: torch.Tensor, take2: torch.Tensor):
        """coper_unpack_rows: (ent_emb[e1] [B, d], ent_emb[e2] [B, d], pred_bias[e2] [B]) out of step 1's gathered buffer: one launch."""
        B, d = int(take1.numel()), self.ent_emb_size
        g1 = torch.empty((B, d), dtype=torch.float32, device=self.device)
        g2 = torch.empty((B, d), dtype=torch.float32, device=self.device)
        b2 = torch.empty((B,), dtype=torch.float32, device=self.device)
        _lib.check(self._h, self._lib.coper_unpack_rows(self._h, _ptr(gathered), _ptr(take1), _ptr(take2), B, _ptr(g1), _ptr(g2), _ptr(b2),
                                                        self._stream()))
        return g1, g2, b2

    def set_x3_ent_absmax(self, absmax: float):
        """coper_set_x3_ent_absmax: the largest |ent_emb| element of the WHOLE table, for a handle that holds a shard of it
        (sharding.py all-reduces it; load_parameters sets it when it is handed the whole table).  No-op in the f32 mode and
        when the value is the one already set; otherwise the handle is prepared again on its next use."""
        if self.score_mode != "bf16x3" or getattr(self, "_x3_absmax", None) == float(absmax):
            return
        _lib.check(self._h, self._lib.coper_set_x3_ent_absmax(self._h, float(absmax)))
        self._x3_absmax = float(absmax)
        self._prepared = False

    def fetch_ranks_audit(self, ranks: torch.Tensor, reset=True):
        """The ranks of the pass just queued (int32 [B], device) AND the band audit's two words through ONE launch into a pinned
        buffer the model keeps (coper_post_ranks_audit), one wait: (ranks int32 ndarray [B] -- a view of that buffer, valid until
        the next call --, audit ratio, audited pairs).  Instead of `.cpu()` (a pageable copy and a wait) followed by
        `band_audit()` (another copy, a memset and a wait): -60 us per call of `ranking_and_hits` on a list of batches."""
        B = int(ranks.numel())
        if ranks.dtype != torch.int32 or not ranks.is_contiguous() or ranks.device != self.device:
            raise ValueError("fetch_ranks_audit: a contiguous int32 tensor on %s" % self.device)
        buf = getattr(self, "_fetch_buf", None)
        if buf is None or buf.numel() < B + 2:
            buf = self._fetch_buf = torch.empty(max(B + 2, 4096), dtype=torch.int32).pin_memory()
        _lib.check(self._h, self._lib.coper_post_ranks_audit(self._h, _ptr(ranks), B, C.c_void_p(buf.data_ptr()), 1 if reset else 0, self._stream()))
        torch.cuda.current_stream(self.device).synchronize()
        out = buf.numpy()
        return out[:B], float(out[B:B + 1].view(np.float32)[0]), int(out[B + 1:B + 2].view(np.uint32)[0])

    def band_audit(self, reset=True):
        """(max |logit_x3 - logit_chain| / (tau / 2), pairs audited) since the last reset (coper_band_audit): the run-time
        check of the bf16x3 mode's exact band.  (0.0, 0) in the f32 mode."""
        r, n = C.c_float(), C.c_int64()
        _lib.check(self._h, self._lib.coper_band_audit(self._h, 1 if reset else 0, C.byref(r), C.byref(n), self._stream()))
        return float(r.value), int(n.value)

    def band_policy(self, ratio, n_pairs):
        """coper_band_policy on the audit's two words: (action, kappa now in force).  action 0 = keep, 1 = the handle's kappa
        was widened for later passes, 2 = widened AND the pass the words belong to must be ranked again."""
        act, kap = C.c_int32(), C.c_float()
        _lib.check(self._h, self._lib.coper_band_policy(self._h, float(ratio), int(n_pairs), C.byref(act), C.byref(kap)))
        return int(act.value), float(kap.value)

    def check_ids(self):
        n = C.c_int64()
        _lib.check(self._h, self._lib.coper_check_ids(self._h, C.byref(n), self._stream()))
        return n.value

    def stale_passes(self):
        """coper_stale_passes: rank passes since the last call that ran on a grouping prepared ahead (`group_next`) whose id arrays had
        been rewritten in between -- their ranks were all written as RANK_STALE (negative) and must be ranked again by a plain call."""
        n = C.c_int64()
        _lib.check(self._h, self._lib.coper_stale_passes(self._h, C.byref(n), self._stream()))
        return n.value

    def profile(self, enable=True):
        _lib.check(self._h, self._lib.coper_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        _lib.check(self._h, self._lib.coper_profile_read(self._h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---------------------------------------------------------------- training
    def train_init(self, seed=0, beta1=0.9, beta2=0.999, epsilon=1e-8, clip_norm=5.0, **overrides):
        """Allocates gradients and AMSGrad slots (amsgrad.py:112-118).  Hyper-parameters come from
        model_descriptors (models.py:99-130) unless overridden.  The tensors passed to load_parameters ARE the
        variables: train_step updates them in place (BN moving statistics included)."""
        md = dict(self.model_descriptors)
        md.update(overrides)
        cfg = _lib.coper_train_config()
        cfg.abi_version = _lib.COPER_ABI_VERSION
        cfg.learning_rate = float(md["learning_rate"])
        cfg.beta1, cfg.beta2, cfg.epsilon, cfg.clip_norm = float(beta1), float(beta2), float(epsilon), float(clip_norm)
        cfg.label_smoothing_epsilon = float(md.get("label_smoothing_epsilon", 0.0))
        cfg.hidden_dropout = float(md.get("hidden_dropout", 0.0))
        cfg.output_dropout = float(md.get("output_dropout", 0.0))
        cfg.context_rel_dropout = float(md.get("context_rel_dropout", 0.0))
        cfg.batch_norm_momentum = float(md.get("batch_norm_momentum", 0.1))
        cfg.batch_norm_train_stats = 1 if md.get("batch_norm_train_stats", False) else 0
        cfg.seed = int(seed) & 0xFFFFFFFF
        with torch.cuda.device(self.device):
            _lib.check(self._h, self._lib.coper_train_init(self._h, C.byref(cfg)))
        self._train_loss = torch.zeros(1, device=self.device, dtype=torch.float32)
        return self

    def train_step(self, batch):
        """`session.run((model.train_op, model.loss))` on one training batch in the reference batch contract
        (models.py:139-152): e1, rel [B]; lookup_values int32 [B,L]; e2_multi float [B,L].  Returns the loss as
        a 1-element device tensor (no synchronisation)."""
        if getattr(self, "_train_loss", None) is None:
            raise _lib.CoperError(5, "call train_init() first")
        e1, rel = self._ids(batch["e1"]), self._ids(batch["rel"])
        lv = batch.get("lookup_values", None)
        one_vs_all = lv is None or (hasattr(lv, "shape") and len(lv.shape) == 2 and lv.shape[1] == 0)   # data.py:322
        lookup = None if one_vs_all else self._ids(lv, torch.int32)
        labels = batch["e2_multi"]
        labels = (labels if isinstance(labels, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(labels)))
        labels = labels.to(device=self.device, dtype=torch.float32).contiguous()
        B, L = labels.shape
        if (lookup is not None and tuple(lookup.shape) != (B, L)) or e1.numel() != B or rel.numel() != B:
            raise ValueError("training batch: e1, rel [B]; lookup_values and e2_multi [B, L] (or e2_multi [B, num_ent] alone)")
        with torch.cuda.device(self.device):
            _lib.check(self._h, self._lib.coper_train_step(self._h, _ptr(e1), _ptr(rel), _ptr(lookup), _ptr(labels), B, L,
                                                           _ptr(self._train_loss), self._stream()))
        self._prepared = False       # caches are stale; the next inference call prepares again
        self._ent_absmax_cache = None
        return self._train_loss

    def train_forward(self, batch, want_predictions=False, want_h=False):
        """The training-mode graph without the update (coper_train_forward): `session.run(model.loss | model.predictions_lookup |
        model.predicted_e2_emb, {model.is_train: True, ...})` when `train_op` is not fetched (models.py:183-192).  Returns
        (loss 1-element device tensor, predictions [B, L] or None, h [B, d] or None); nothing is updated."""
        if getattr(self, "_train_loss", None) is None:
            self.train_init()
        e1, rel = self._ids(batch["e1"]), self._ids(batch["rel"])
        lv = batch.get("lookup_values", None)
        one_vs_all = lv is None or (hasattr(lv, "shape") and len(lv.shape) == 2 and lv.shape[1] == 0)
        lookup = None if one_vs_all else self._ids(lv, torch.int32)
        labels = batch["e2_multi"]
        labels = (labels if isinstance(labels, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(labels)))
        labels = labels.to(device=self.device, dtype=torch.float32).contiguous()
        B, L = labels.shape
        if (lookup is not None and tuple(lookup.shape) != (B, L)) or e1.numel() != B or rel.numel() != B:
            raise ValueError("training batch: e1, rel [B]; lookup_values and e2_multi [B, L] (or e2_multi [B, num_ent] alone)")
        loss = torch.zeros(1, device=self.device, dtype=torch.float32)
        pred = torch.empty((B, L), device=self.device, dtype=torch.float32) if want_predictions else None
        hv = torch.empty((B, self.ent_emb_size), device=self.device, dtype=torch.float32) if want_h else None
        with torch.cuda.device(self.device):
            _lib.check(self._h, self._lib.coper_train_forward(self._h, _ptr(e1), _ptr(rel), _ptr(lookup), _ptr(labels), B, L, _ptr(loss),
                                                              _ptr(pred), _ptr(hv), self._stream()))
        return loss, pred, hv

    def train_grad(self, leaf_name):
        """(gradient of the last step as a tensor copy, global gradient norm) -- test / diagnostics hook."""
        n, gn = C.c_int64(), C.c_double()
        _lib.check(self._h, self._lib.coper_train_grad(self._h, leaf_name.encode(), None, 0, C.byref(n), None, self._stream()))
        out = torch.empty(n.value, device=self.device, dtype=torch.float32)
        _lib.check(self._h, self._lib.coper_train_grad(self._h, leaf_name.encode(), _ptr(out), n.value, C.byref(n), C.byref(gn),
                                                       self._stream()))
        return out, gn.value

    def optimizer_state(self):
        """AMSGrad state for a checkpoint: ({leaf: (m, v, v_hat) numpy}, {"beta1_power", "beta2_power", "step"}) --
        what `tf.train.Saver` keeps as `<var>/AMSGrad{,_1,_2}` and beta1_power / beta2_power (amsgrad.py:108-119)."""
        slots = {}
        for leaf in self.trainable_leaves():
            parts = []
            for which in range(3):
                n = C.c_int64()
                _lib.check(self._h, self._lib.coper_train_slot(self._h, leaf.encode(), which, None, 0, 0, C.byref(n), self._stream()))
                buf = torch.empty(n.value, device=self.device, dtype=torch.float32)
                _lib.check(self._h, self._lib.coper_train_slot(self._h, leaf.encode(), which, _ptr(buf), n.value, 0, C.byref(n),
                                                               self._stream()))
                parts.append(buf.cpu().numpy().reshape(tuple(self._tensors[leaf].shape)))
            slots[leaf] = tuple(parts)
        b1, b2, st = C.c_double(), C.c_double(), C.c_int64()
        _lib.check(self._h, self._lib.coper_train_powers(self._h, None, None, None, C.byref(b1), C.byref(b2), C.byref(st)))
        return slots, dict(beta1_power=b1.value, beta2_power=b2.value, step=st.value)

    def load_optimizer_state(self, slots, powers=None):
        """Inverse of optimizer_state (after train_init): resume training from a checkpoint's slots."""
        for leaf, parts in slots.items():
            for which, a in enumerate(parts):
                if a is None:
                    continue
                src = torch.as_tensor(np.ascontiguousarray(np.asarray(a, np.float32))).reshape(-1).to(self.device)
                n = C.c_int64()
                _lib.check(self._h, self._lib.coper_train_slot(self._h, leaf.encode(), which, _ptr(src), src.numel(), 1, C.byref(n),
                                                               self._stream()))
                torch.cuda.current_stream(self.device).synchronize()    # src may be freed after this
        if powers:
            b1 = C.c_double(powers["beta1_power"]) if "beta1_power" in powers else None
            b2 = C.c_double(powers["beta2_power"]) if "beta2_power" in powers else None
            st = C.c_int64(int(powers["step"])) if "step" in powers else None
            _lib.check(self._h, self._lib.coper_train_powers(self._h, C.byref(b1) if b1 else None, C.byref(b2) if b2 else None,
                                                            C.byref(st) if st else None, None, None, None))
        return self

    def trainable_leaves(self):
        """Leaves with optimizer slots (everything but the BN moving statistics)."""
        return [k for k in self._tensors if not k.endswith(("moving_mean", "moving_variance"))]

    # ---------------------------------------------------------------- reference-style access
    @property
    def last_loss(self):
        """The last training step's loss (a 1-element device tensor); `session.run(model.loss, ...)` returns it as a float."""
        if getattr(self, "_train_loss", None) is None:
            raise NotImplementedError("no training step has run: call train_init() and train_step(batch)")
        return self._train_loss

    def session(self):
        return Session(self)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            st = getattr(self, "_stage", None)
            if st is not None:                       # the staging launches may still be reading the pinned buffers
                for ev in st["ev"]:
                    ev.synchronize()
                self._stage = None
            self._lib.coper_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Session(object):
    """`session.run(fetches, feed_dict)` shim: `feed_dict[model.input_iterator_handle]` is any iterator
    over batches in the reference batch contract (see coper_amd.data).  Each `run` consumes one batch
    (like one `session.run` on the TF iterator, metrics.py:40-42) and returns host NumPy arrays."""

    def __init__(self, model: ConvE):
        self.model = model
        self._iters = {}

    def run(self, fetches, feed_dict=None):
        m = self.model
        feed_dict = feed_dict or {}
        single = not isinstance(fetches, (tuple, list))
        fl = [fetches] if single else list(fetches)
        # variables fetched directly, `session.run([model.variables['rel_emb'], model.variables['ent_emb']])`
        # (run_cpg.py:244-248): device tensors -> host arrays; no batch is consumed
        if all(isinstance(f, torch.Tensor) for f in fl):
            out = [f.detach().cpu().numpy() for f in fl]
            return out[0] if single else (out if isinstance(fetches, list) else tuple(out))
        handle = feed_dict.get(m.input_iterator_handle)
        if handle is None:
            raise ValueError("feed_dict must map model.input_iterator_handle to a batch iterator")
        key = id(handle)
        if key not in self._iters:
            self._iters[key] = iter(handle)
        try:
            batch = next(self._iters[key])
        except StopIteration:
            del self._iters[key]
            raise OutOfRangeError()
        is_train = bool(feed_dict.get(m.is_train, False))
        names = [getattr(f, "name", None) for f in fl]
        if "train_op" in names:
            # one optimisation step on this batch (run_cpg.py:211-219: `loss, _ = session.run((model.loss, model.train_op), feed)`);
            # TF evaluates `loss` and the update in one graph execution: the loss returned is the loss of THIS batch before
            # the update, which is what coper_train_step reports
            if not is_train:
                raise ValueError("train_op needs feed_dict[model.is_train] = True (run_cpg.py:212)")
            if getattr(m, "_train_loss", None) is None:
                m.train_init()
            loss = m.train_step(batch)
            out = []
            for f, n in zip(fl, names):
                if n == "train_op":
                    out.append(None)
                elif n == "loss":
                    out.append(float(loss.cpu()[0]))
                elif n in ("e1", "e2", "rel", "e2_multi", "lookup_values"):
                    out.append(_host(batch[n]))      # DeviceTrainDataset yields device tensors
                elif n is None and f is None:       # model.summaries is None in this build (run_cpg.py:216 checks for it)
                    out.append(None)
                else:
                    raise KeyError("fetch %r cannot be combined with train_op" % (n,))
            return out[0] if single else tuple(out)
        if is_train:
            # the training-mode graph without train_op (models.py:183-192): dropout and batch statistics, nothing updated
            loss, pred, hv = m.train_forward(batch, want_predictions=any(n in ("predictions_lookup", "predictions_all") for n in names),
                                             want_h="predicted_e2_emb" in names)
            lv = batch.get("lookup_values", None)
            one_vs_all = lv is None or (hasattr(lv, "shape") and len(lv.shape) == 2 and lv.shape[1] == 0)
            out = []
            for f, n in zip(fl, names):
                if n == "loss":
                    out.append(float(loss.cpu()[0]))
                elif n == "predictions_lookup" or (n == "predictions_all" and one_vs_all):
                    out.append(pred.cpu().numpy())
                elif n == "predicted_e2_emb":
                    out.append(hv.cpu().numpy())
                elif n in ("e1", "e2", "rel", "e2_multi", "lookup_values"):
                    out.append(_host(batch[n]))
                elif n is None and f is None:
                    out.append(None)
                else:
                    raise KeyError("fetch %r is not served under is_train=True without train_op" % (n,))
            return out[0] if single else tuple(out)
        cache = {}

        def h():
            if "h" not in cache:
                cache["h"] = m.encode(batch["e1"], batch["rel"])
            return cache["h"]

        out = []
        for f in fl:
            n = f.name
            if n in ("e1", "e2", "rel"):
                out.append(_host(batch[n]))
            elif n == "e2_multi":
                if "e2_multi" in batch:
                    out.append(_host(batch["e2_multi"]))
                else:
                    from .data import csr_to_dense_filter
                    out.append(csr_to_dense_filter(batch["filt_indptr"], batch["filt_idx"], m.num_ent))
            elif n == "lookup_values":
                out.append(_host(batch["lookup_values"]))
            elif n == "predicted_e2_emb":
                out.append(h().cpu().numpy())
            elif n == "predictions_all":
                out.append(m.score_all(h()).cpu().numpy())
            elif n == "predictions_lookup":
                out.append(m.score_lookup(h(), batch["lookup_values"]).cpu().numpy())
            elif n == "loss":
                out.append(float(m.last_loss.cpu()[0]))
            else:
                raise KeyError(n)
        return out[0] if single else tuple(out)
