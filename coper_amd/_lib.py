"""ctypes binding of libcoper_hip.so (include/coper_hip.h).

This is the "reference-side binding" of INTEGRATION.md: plain pointers and sizes, no torch
types in any signature.  There is NO CPU fallback: if the shared library is missing or
cannot be loaded, every entry point raises (``CoperLibraryError``)."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcoper_hip.so")

COPER_ABI_VERSION = 3
COPER_MAX_CTX = 8

SCORE_F32, SCORE_BF16X3 = 0, 1
RANK_STALE = -(1 << 30)      # COPER_RANK_STALE (include/coper_hip.h)

STATUS = {0: "COPER_OK", 1: "COPER_EINVAL", 2: "COPER_EMISSING", 3: "COPER_ESHAPE", 4: "COPER_EHIP",
          5: "COPER_ESTATE", 6: "COPER_ENOMEM", 7: "COPER_EUNSUPPORTED"}


class CoperLibraryError(RuntimeError):
    pass


class CoperError(RuntimeError):
    def __init__(self, code, text):
        self.code = code
        RuntimeError.__init__(self, "%s: %s" % (STATUS.get(code, str(code)), text))


class coper_config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32),
        ("num_ent", C.c_int64), ("num_rel", C.c_int64),
        ("ent_emb_size", C.c_int32), ("rel_emb_size", C.c_int32),
        ("emb_h", C.c_int32), ("emb_w", C.c_int32),
        ("conv_filter_height", C.c_int32), ("conv_filter_width", C.c_int32), ("conv_num_channels", C.c_int32),
        ("concat_rel", C.c_int32), ("do_parameter_lookup", C.c_int32),
        ("n_ctx_conv", C.c_int32), ("ctx_conv", C.c_int32 * COPER_MAX_CTX),
        ("n_ctx_out", C.c_int32), ("ctx_out", C.c_int32 * COPER_MAX_CTX),
        ("context_rel_use_batch_norm", C.c_int32), ("bn_epsilon", C.c_float),
        ("shard_lo", C.c_int64), ("shard_hi", C.c_int64),
        ("score_mode", C.c_int32), ("rank_band_kappa", C.c_float), ("x3_ent_absmax", C.c_float),
        ("band_audit_period", C.c_int32), ("role", C.c_int32), ("rel_mod_world", C.c_int32), ("rel_mod_rank", C.c_int32), ("reserved", C.c_int32 * 1),
    ]


_P = C.c_void_p
_I64 = C.c_int64

class coper_train_config(C.Structure):
    """include/coper_hip.h: coper_train_config."""
    _fields_ = [("abi_version", C.c_int32), ("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float), ("clip_norm", C.c_float), ("label_smoothing_epsilon", C.c_float),
                ("hidden_dropout", C.c_float), ("output_dropout", C.c_float), ("batch_norm_momentum", C.c_float),
                ("batch_norm_train_stats", C.c_int32), ("seed", C.c_uint32), ("context_rel_dropout", C.c_float),
                ("reserved", C.c_int32 * 7)]


# name -> (restype, argtypes): every symbol include/coper_hip.h declares
PROTOTYPES = {
    "coper_abi_version": (C.c_int, []),
    "coper_crc32c": (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_uint64]),
    "coper_create": (C.c_int, [C.POINTER(coper_config), C.POINTER(_P)]),
    "coper_destroy": (None, [_P]),
    "coper_last_error": (C.c_char_p, [_P]),
    "coper_get_dims": (C.c_int, [_P, C.POINTER(_I64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(_I64)]),
    "coper_num_params": (C.c_int, [_P]),
    "coper_param_spec": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(_I64), C.POINTER(C.c_int)]),
    "coper_set_param": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(_I64), C.c_int]),
    "coper_set_x3_ent_absmax": (C.c_int, [_P, C.c_float]),
    "coper_band_audit": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_float), C.POINTER(_I64), _P]),
    "coper_band_audit_post": (C.c_int, [_P, C.c_int32, _P, _P]),
    "coper_post_ranks_audit": (C.c_int, [_P, _P, _I64, _P, C.c_int32, _P]),
    "coper_pack_owned_rows": (C.c_int, [_P, _P, _I64, _I64, C.c_float, C.c_float, _P, _P]),
    "coper_unpack_rows": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "coper_pack_shard_record": (C.c_int, [_P, _P, _P, _P, _P, _I64, C.c_int32, C.c_int32, _P, _P]),
    "coper_merge_shard_records": (C.c_int, [_P, _P, C.c_int32, _I64, C.c_int32, _P, _P, _P, _P, _P]),
    "coper_pack_ids_i32": (C.c_int, [_P, _I64, _P, _P, _I64, _P]),
    "coper_hits_means": (C.c_int, [_P, _I64, _P, C.c_int32, _P, _P, _P]),
    "coper_sample_train_batch": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, C.c_int32, C.c_double, _I64, C.c_uint64,
                                           C.c_uint64, _P, _P, _P, _P, _P, _P]),
    "coper_band_policy": (C.c_int, [_P, C.c_float, _I64, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "coper_prepare": (C.c_int, [_P, _P]),
    "coper_reserve": (C.c_int, [_P, _I64, _I64, _P]),
    "coper_widen_ids": (C.c_int, [_P, _P, _I64, _P, _P]),
    "coper_copy_out_i32": (C.c_int, [_P, _P, _I64, _P, _P]),
    "coper_stage_ids_next": (C.c_int, [_P, _P, _I64, _P]),
    "coper_post_i32_next": (C.c_int, [_P, _P, _I64, _P]),
    "coper_group_next": (C.c_int, [_P, _P, _P, _I64, C.c_int32]),
    "coper_gather_entities": (C.c_int, [_P, _P, _I64, _P, _P]),
    "coper_encode": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P]),
    "coper_score_all": (C.c_int, [_P, _P, _I64, _P, _I64, _P]),
    "coper_score_lookup": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P]),
    "coper_target_scores": (C.c_int, [_P, _P, _P, _I64, _P, _P]),
    "coper_score_rows": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P]),
    "coper_rank_counts": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, C.c_int32, _P, _P, _P, _P, _P]),
    "coper_rank": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "coper_encode_rank": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P]),
    "coper_check_ids": (C.c_int, [_P, C.POINTER(_I64), _P]),
    "coper_stale_passes": (C.c_int, [_P, C.POINTER(_I64), _P]),
    "coper_live_device_bytes": (_I64, []),
    "coper_profile_enable": (C.c_int, [_P, C.c_int]),
    "coper_profile_read": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_I64)]),
    "coper_train_init": (C.c_int, [_P, C.POINTER(coper_train_config)]),
    "coper_train_step": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "coper_train_forward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P]),
    "coper_train_grad": (C.c_int, [_P, C.c_char_p, _P, _I64, C.POINTER(_I64), C.POINTER(C.c_double), _P]),
    "coper_train_slot": (C.c_int, [_P, C.c_char_p, C.c_int32, _P, _I64, C.c_int32, C.POINTER(_I64), _P]),
    "coper_train_powers": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(_I64)]),
}

_lib = None


def load(path=None):
    """Loads the shared library (once).  `import torch` must already have happened in a
    process that also uses torch on the GPU, so both share one HIP runtime (libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("COPER_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise CoperLibraryError(
            "libcoper_hip.so not found at %s -- build it with `python -m coper_amd.build` "
            "(there is no CPU fallback for the product path)" % path)
    try:
        lib = C.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise CoperLibraryError("cannot load %s: %s" % (path, e))
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise CoperLibraryError("%s does not export %s" % (path, name))
        fn.restype = res
        fn.argtypes = args
    if lib.coper_abi_version() != COPER_ABI_VERSION:
        raise CoperLibraryError("ABI version mismatch: library %d, binding %d" % (lib.coper_abi_version(), COPER_ABI_VERSION))
    _lib = lib
    return lib


def check(handle, rc):
    if rc != 0:
        lib = load()
        text = lib.coper_last_error(handle)
        raise CoperError(rc, text.decode() if text else "")


ROLE_BOTH, ROLE_ENCODE, ROLE_SCORE = 0, 1, 2     # coper_role


def make_config(md, device=0, shard=None, score_mode=SCORE_F32, bn_epsilon=1e-3, rank_band_kappa=0.0, x3_ent_absmax=0.0,
                band_audit_period=0, role=ROLE_BOTH, rel_mod=None):
    """model_descriptors dict (models.py:98-130 keys) -> coper_config."""
    cfg = coper_config()
    cfg.abi_version = COPER_ABI_VERSION
    cfg.device = int(device)
    cfg.num_ent = int(md["num_ent"])
    cfg.num_rel = int(md["num_rel"])
    d = int(md["ent_emb_size"])
    cfg.ent_emb_size = d
    cfg.rel_emb_size = int(md["rel_emb_size"])
    emb_h = int(md.get("emb_h", 10))
    cfg.emb_h = emb_h
    cfg.emb_w = int(md.get("emb_w", d // emb_h))
    cfg.conv_filter_height = int(md.get("conv_filter_height", 3))
    cfg.conv_filter_width = int(md.get("conv_filter_width", 3))
    cfg.conv_num_channels = int(md.get("conv_num_channels", 32))
    cfg.concat_rel = 1 if md.get("concat_rel", False) else 0
    cfg.do_parameter_lookup = 1 if md.get("do_parameter_lookup", False) else 0
    for key, nfield, afield in (("context_rel_conv", "n_ctx_conv", "ctx_conv"), ("context_rel_out", "n_ctx_out", "ctx_out")):
        ctx = md.get(key, None)
        if ctx is None:
            setattr(cfg, nfield, -1)
        else:
            ctx = [int(v) for v in ctx]
            if len(ctx) > COPER_MAX_CTX:
                raise ValueError("%s: at most %d hidden layers" % (key, COPER_MAX_CTX))
            setattr(cfg, nfield, len(ctx))
            arr = getattr(cfg, afield)
            for i, v in enumerate(ctx):
                arr[i] = v
    cfg.context_rel_use_batch_norm = 1 if md.get("context_rel_use_batch_norm", False) else 0
    cfg.bn_epsilon = float(bn_epsilon)
    lo, hi = shard if shard is not None else (0, cfg.num_ent)
    cfg.shard_lo, cfg.shard_hi = int(lo), int(hi)
    cfg.score_mode = int(score_mode)
    cfg.rank_band_kappa = float(rank_band_kappa)     # 0: the library default (include/coper_hip.h)
    cfg.x3_ent_absmax = float(x3_ent_absmax)         # 0: the handle's own rows; entity shards pass the table-wide maximum
    cfg.band_audit_period = int(band_audit_period)   # 0: the library default (first count launch, then every 8th)
    cfg.role = int(role)                             # 0: encoder and scorer (every handle before round 6)
    if rel_mod is not None:                          # (G, g): generated dense weights of the relations r with r mod G == g only
        cfg.rel_mod_world, cfg.rel_mod_rank = int(rel_mod[0]), int(rel_mod[1])
    return cfg
