"""CPU: the C-ABI library loads, exports every symbol include/coper_hip.h declares, validates
configurations like the reference derives its shapes, and fails loudly without a GPU.
No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from coper_amd import _lib, data as cdata

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "coper_hip.h")).read()
    return sorted(set(re.findall(r"^COPER_API [^;(]*?\b(coper_[a-z0-9_]+)\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.PROTOTYPES) == names          # the ctypes binding covers the whole header
    assert lib.coper_abi_version() == _lib.COPER_ABI_VERSION


def _create(md, **kw):
    lib = _lib.load()
    cfg = _lib.make_config(md, **kw)
    h = C.c_void_p()
    rc = lib.coper_create(C.byref(cfg), C.byref(h))
    return lib, rc, h


@pytest.mark.parametrize("name", ["nations_cpg", "fb15k237_cpg", "wn18rr_cpg", "fb15k237_plain", "synth10m_cpg"])
def test_param_specs_match_reference_shapes(name):
    md = cdata.model_descriptors(name)
    lib, rc, h = _create(md)
    assert rc == 0, lib.coper_last_error(None)
    want = cdata.param_shapes(md)
    got = {}
    for i in range(lib.coper_num_params(h)):
        nm, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        assert lib.coper_param_spec(h, i, C.byref(nm), shape, C.byref(nd)) == 0
        got[nm.value.decode()] = tuple(shape[j] for j in range(nd.value))
    assert got == {k: tuple(v) for k, v in want.items()}
    F, Ho, Wo, nl = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int64()
    lib.coper_get_dims(h, C.byref(F), C.byref(Ho), C.byref(Wo), C.byref(nl))
    exp = {"nations_cpg": (384, 2, 6), "fb15k237_cpg": (4608, 8, 18), "wn18rr_cpg": (4608, 8, 18),
           "fb15k237_plain": (10368, 18, 18), "synth10m_cpg": (6272, 14, 14)}[name]
    assert (F.value, Ho.value, Wo.value) == exp and nl.value == md["num_ent"]
    lib.coper_destroy(h)


def test_generator_mlp_and_lookup_specs():
    md = cdata.model_descriptors("fb15k237_cpg", context_rel_out=[64], context_rel_conv=[16])
    lib, rc, h = _create(md)
    assert rc == 0
    names = set()
    for i in range(lib.coper_num_params(h)):
        nm = C.c_char_p()
        lib.coper_param_spec(h, i, C.byref(nm), None, None)
        names.add(nm.value.decode())
    assert names == set(cdata.param_shapes(md))
    assert "fc_weights/CPG/Projection0/BatchNorm/moving_variance" in names
    lib.coper_destroy(h)
    md = cdata.model_descriptors("fb15k237_cpg", do_parameter_lookup=True, context_rel_conv=[])
    lib, rc, h = _create(md)
    assert rc == 0
    lib.coper_destroy(h)


@pytest.mark.parametrize("over,frag", [
    (dict(emb_h=7), "emb_h * emb_w"),
    (dict(context_rel_out=None, rel_emb_size=32), "plain ConvE stacks"),
    (dict(context_rel_out=None, context_rel_conv=None, do_parameter_lookup=True, rel_emb_size=200), "ill-formed"),
    (dict(do_parameter_lookup=True, concat_rel=True), "concat_rel"),
    (dict(num_rel=0), "positive"),
    (dict(conv_filter_height=11), "larger than the image"),
])
def test_ill_formed_configs_are_rejected(over, frag):
    md = cdata.model_descriptors("fb15k237_cpg", **over)
    lib, rc, h = _create(md)
    assert rc == 1 and not h.value
    assert frag in lib.coper_last_error(None).decode()


def test_bad_shard_and_state_errors():
    md = cdata.model_descriptors("nations_cpg")
    lib, rc, h = _create(md, shard=(5, 3))
    assert rc == 1
    lib, rc, h = _create(md, shard=(0, 7))
    assert rc == 0
    nl = C.c_int64()
    lib.coper_get_dims(h, None, None, None, C.byref(nl))
    assert nl.value == 7
    # call-order / argument errors that need no device
    assert lib.coper_encode(h, None, None, 4, None, None, None) == 5          # ESTATE: not prepared
    assert b"coper_prepare" in lib.coper_last_error(h)
    bogus = (C.c_int64 * 1)(3)
    assert lib.coper_set_param(h, b"no_such_param", C.c_void_p(16), bogus, 1) == 1
    assert lib.coper_set_param(h, b"pred_bias", C.c_void_p(16), bogus, 1) == 3   # ESHAPE: needs 7
    assert lib.coper_prepare(h, None) == 2                                       # EMISSING
    assert b"never set" in lib.coper_last_error(h)
    lib.coper_destroy(h)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.CoperLibraryError):
        _lib.load(str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    _lib.load()


def test_model_without_gpu_raises():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from coper_amd.models import ConvE
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ConvE(cdata.model_descriptors("nations_cpg"))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "coper_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "coper_oracle" not in text, f
