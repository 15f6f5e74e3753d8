"""TF `Saver` checkpoint reader / writer (run_cpg.py:189,206,252).  No TensorFlow here, so the
format pieces with published known answers are pinned to those, and the rest by the round trip."""
import os
import struct

import numpy as np
import pytest

from coper_amd import data as cdata
from coper_amd import tf_bundle as tb
from coper_amd import weights


def test_crc32c_known_answers():
    # RFC 3720 B.4 / leveldb crc32c_test.cc
    assert tb.crc32c(b"123456789") == 0xE3069283
    assert tb.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert tb.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E
    assert tb.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert tb.crc32c(b"") == 0
    # the mask is an invertible rotation + offset (leveldb crc32c.h)
    c = tb.crc32c(b"foo")
    m = tb.masked_crc32c(b"foo")
    assert m != c and (((m - 0xA282EAD8) & 0xFFFFFFFF) >> 17 | ((m - 0xA282EAD8) & 0xFFFFFFFF) << 15) & 0xFFFFFFFF == c


def test_crc32c_lane_path_equals_scalar_path():
    rng = np.random.default_rng(0)
    for n in (64 * 4096, 64 * 4096 + 1, 1_000_003):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert tb.crc32c(b, native=False) == tb._crc_update_scalar(0xFFFFFFFF, b) ^ 0xFFFFFFFF


def test_native_crc32c_equals_the_numpy_form(monkeypatch):
    """coper_crc32c of libcoper_hip.so (host code: crc32 instruction, or slicing-by-8 tables) against the NumPy form and the
    RFC 3720 known answers; chaining over pieces; unaligned starts."""
    import ctypes
    from coper_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(1)
    assert lib.coper_crc32c(0, b"123456789", 9) == 0xE3069283
    assert lib.coper_crc32c(0, bytes(32), 32) == 0x8A9136AA and lib.coper_crc32c(0, b"", 0) == 0
    for n in (1, 7, 8, 9, 63, 4096, 64 * 4096 + 5, 1_000_003):
        b = rng.integers(0, 256, n + 3, dtype=np.uint8)
        for off in (0, 1, 3):
            piece = b[off:off + n]
            want = tb.crc32c(piece.tobytes(), native=False)
            assert lib.coper_crc32c(0, piece.ctypes.data, n) == want, (n, off)
            if n >= 4096:
                assert tb.crc32c(piece.tobytes()) == want                       # the route tf_bundle takes by default
        k = n // 3
        c = lib.coper_crc32c(0, b.ctypes.data, k)
        assert lib.coper_crc32c(c, b.ctypes.data + k, n - k) == tb.crc32c(b[:n].tobytes(), native=False)


def test_varint_and_snappy():
    for v in (0, 1, 127, 128, 300, 2**32 - 1, 2**63 - 1):
        enc = tb._put_varint(v)
        assert tb._get_varint(enc, 0) == (v, len(enc))
    assert tb._put_varint(300) == b"\xac\x02"                      # protobuf encoding guide's example
    # raw snappy: length 11, literal "abc" (tag (3-1)<<2), copy-1 len 4 off 3, literal "xy", copy-2 len 2 off 2
    comp = bytes([11, (3 - 1) << 2]) + b"abc" + bytes([((4 - 4) << 2) | 1 | (0 << 5), 3]) + bytes([(2 - 1) << 2]) + b"xy" + \
        bytes([((2 - 1) << 2) | 2, 2, 0])
    assert tb._snappy_uncompress(comp) == b"abcabcaxyxy"


def _params():
    md = dict(cdata._COMMON)
    md.update(num_ent=300, num_rel=6, ent_emb_size=40, rel_emb_size=8, emb_h=10, emb_w=4, conv_num_channels=8,
              context_rel_conv=[5], context_rel_out=[7], context_rel_use_batch_norm=True)
    return cdata.synthetic_params(md, seed=2)


def test_checkpoint_round_trip_and_names(tmp_path):
    p = {k: np.asarray(v, np.float32) for k, v in _params().items()}
    prefix = str(tmp_path / "ckpt" / "model_weights.ckpt")
    weights.save_tf_checkpoint(prefix, p)
    listing = tb.list_bundle(prefix)
    # the reference's variable names
    assert "variables/variables/ent_emb" in listing and "variables/Conv1BN/moving_mean" in listing
    assert "variables/variables/fc_weights/CPG/Projection0" in listing
    assert "variables/fc_weights/CPG/Projection0/BatchNorm/gamma" in listing
    assert list(listing) == sorted(listing, key=lambda s: s.encode())       # table order
    e = listing["variables/variables/ent_emb"]
    assert e["shape"] == (300, 40) and e["dtype"] == np.float32 and e["size"] == 300 * 40 * 4
    back = weights.load_tf_checkpoint(prefix)
    assert sorted(back) == sorted(p)
    for k in p:
        assert back[k].dtype == np.float32 and np.array_equal(back[k], p[k]), k
    # the index is a leveldb table: magic in the last 8 bytes, footer 48 bytes
    idx = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", idx[-8:])[0] == 0xDB4775248B80FB57


def test_checkpoint_with_optimizer_slots_scalars_and_many_blocks(tmp_path):
    rng = np.random.default_rng(1)
    t = {}
    for i in range(300):                                   # several data blocks, prefix-compressed keys
        t["variables/variables/layer_%03d/kernel" % i] = rng.normal(size=(3, i % 5 + 1)).astype(np.float32)
    t["variables/variables/ent_emb"] = rng.normal(size=(50, 8)).astype(np.float32)
    for suf in ("AMSGrad", "AMSGrad_1", "AMSGrad_2"):
        t["variables/variables/ent_emb/" + suf] = rng.normal(size=(50, 8)).astype(np.float32)
    t["variables/beta1_power"] = np.float32(0.81)
    t["variables/beta2_power"] = np.float32(0.998)
    t["variables/global_step"] = np.int64(1234)
    prefix = str(tmp_path / "m.ckpt")
    tb.write_bundle(prefix, t, block_size=512)
    raw = tb.read_bundle(prefix)
    assert sorted(raw) == sorted(t)
    for k in t:
        assert raw[k].dtype == np.asarray(t[k]).dtype and np.array_equal(raw[k], t[k]), k
    assert raw["variables/global_step"].shape == () and int(raw["variables/global_step"]) == 1234
    params, slots, powers = weights.load_tf_checkpoint(prefix, with_optimizer=True)
    assert "ent_emb" in params and "global_step" not in params and len(params) == 301
    m, v, vh = slots["ent_emb"]
    assert np.array_equal(m, t["variables/variables/ent_emb/AMSGrad"]) and np.array_equal(vh, t["variables/variables/ent_emb/AMSGrad_2"])
    assert abs(powers["beta1_power"] - 0.81) < 1e-6
    assert list(tb.read_bundle(prefix, names=["variables/beta1_power"])) == ["variables/beta1_power"]
    with pytest.raises(KeyError):
        tb.read_bundle(prefix, names=["nope"])


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "c.ckpt")
    tb.write_bundle(prefix, {"a": np.arange(100, dtype=np.float32), "b": np.ones((4, 4), np.float32)})
    data_file = prefix + ".data-00000-of-00001"
    blob = bytearray(open(data_file, "rb").read())
    blob[17] ^= 0x40
    open(data_file, "wb").write(bytes(blob))
    with pytest.raises(ValueError, match="data checksum"):
        tb.read_bundle(prefix)
    assert np.array_equal(tb.read_bundle(prefix, names=["b"])["b"], np.ones((4, 4), np.float32))
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[5] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="checksum"):
        tb.list_bundle(prefix)
    open(prefix + ".index", "wb").write(b"not a table")
    with pytest.raises(ValueError, match="footer|magic"):
        tb.list_bundle(prefix)
    open(data_file, "wb").write(bytes(blob[:100]))
    tb.write_bundle(prefix + "2", {"a": np.arange(100, dtype=np.float32)})
    open(prefix + "2.data-00000-of-00001", "wb").write(b"\0" * 10)
    with pytest.raises(ValueError, match="bytes on disk"):
        tb.read_bundle(prefix + "2")


def test_reader_on_a_hand_assembled_checkpoint(tmp_path):
    """The reader against bytes the writer of this package did NOT produce: tests/golden/tf_bundle_handmade.json holds
    a two-block table assembled byte by byte from the published table / proto layouts by
    tests/golden/make_tf_bundle_handmade.py (own bitwise CRC-32C, restart points inside a block, a prefix-compressed key,
    a scalar int64 tensor, omitted zero fields, a shortened index separator).  Not a TensorFlow-written file: none exists
    in this environment, the format stays restated-from-publication."""
    import json
    from coper_amd import tf_bundle
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_bundle_handmade.json")))
    prefix = str(tmp_path / "model.ckpt")
    open(prefix + ".index", "wb").write(bytes.fromhex(g["index_hex"]))
    open(prefix + ".data-00000-of-00001", "wb").write(bytes.fromhex(g["data_hex"]))
    listed = tf_bundle.list_bundle(prefix)
    assert sorted(listed) == sorted(g["tensors"])
    got = tf_bundle.read_bundle(prefix)
    for name, want in g["tensors"].items():
        a = got[name]
        assert a.dtype == np.dtype(want["dtype"]) and list(a.shape) == want["shape"], name
        assert np.array_equal(a.reshape(-1), np.asarray(want["values"], a.dtype)), name
    # a flipped tensor byte and a flipped table byte are both detected by the stored CRCs
    data = bytearray(bytes.fromhex(g["data_hex"]))
    data[9] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(Exception):
        tf_bundle.read_bundle(prefix)
    open(prefix + ".data-00000-of-00001", "wb").write(bytes.fromhex(g["data_hex"]))
    index = bytearray(bytes.fromhex(g["index_hex"]))
    index[20] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(index))
    with pytest.raises(Exception):
        tf_bundle.read_bundle(prefix)
