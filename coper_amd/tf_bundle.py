"""Reader / writer for the TensorFlow `Saver` (V2) checkpoint the reference driver keeps
(`run_cpg.py:92-94,189,206,252`: `tf.train.Saver().save(session, '.../model_weights.ckpt')`), SURVEY.md 8f-3.

TensorFlow is not available in this environment, so the format is restated from its published layout
(tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/{table,block,format}; TF 1.14 is what the reference
pins, `CoPER_ConvE/README.md:115-116`) -- there is no TF-written file here to pin it against; the tests pin the
pieces that have published known answers (CRC-32C check values, the table magic, varint / protobuf wire rules)
and the round trip through the writer below.

  <prefix>.index                 an immutable sorted string table ("leveldb table"):
      data blocks   entries  varint32 shared | varint32 non_shared | varint32 value_len | key suffix | value,
                    then fixed32 restart offsets, fixed32 num_restarts
      block trailer 1 byte compression (0 none, 1 snappy) + fixed32 masked CRC-32C of (block | type byte)
      index block   last-key separator -> BlockHandle (varint64 offset, varint64 size)
      footer        48 bytes: metaindex handle, index handle, zero padding to 40, magic 0xdb4775248b80fb57 (LE)
    key ""   -> BundleHeaderProto  {1: num_shards, 2: endianness (0 little), 3: VersionDef {1: producer}}
    key name -> BundleEntryProto   {1: dtype, 2: TensorShapeProto {2: Dim {1: size}}, 3: shard_id, 4: offset,
                                    5: size, 6: fixed32 masked crc32c of the tensor bytes, 7: slices}
  <prefix>.data-SSSSS-of-NNNNN   tensor bytes (row-major, little endian) at [offset, offset + size) of shard SSSSS
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterable, Tuple

import numpy as np

__all__ = ["read_bundle", "write_bundle", "list_bundle", "crc32c", "masked_crc32c"]

TABLE_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"),
           6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("bool"), 17: np.dtype("<u2"), 19: np.dtype("<f2"),
           22: np.dtype("<u4"), 23: np.dtype("<u8")}
_DTYPE_IDS = {v: k for k, v in _DTYPES.items()}

# ---------------------------------------------------------------------------------------------------------
# CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), masked as leveldb / TF store it


def _make_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        t[i] = c
    return t


_T = _make_table()
_TL = [int(x) for x in _T]


def _crc_update_scalar(state: int, data: bytes) -> int:
    for b in data:
        state = _TL[(state ^ b) & 0xFF] ^ (state >> 8)
    return state


def _zero_shift_matrix(nbytes: int):
    """32 columns of the GF(2) map state -> state after feeding `nbytes` zero bytes (for combining lane CRCs)."""
    def apply(cols, x):
        y = 0
        i = 0
        while x:
            if x & 1:
                y ^= cols[i]
            x >>= 1
            i += 1
        return y

    one = [_crc_update_scalar(1 << i, b"\0") for i in range(32)]      # one zero byte
    result = [1 << i for i in range(32)]                               # identity
    sq, n = one, nbytes
    while n:
        if n & 1:
            result = [apply(sq, c) for c in result]
        sq = [apply(sq, c) for c in sq]
        n >>= 1
    return result


_native = None   # coper_crc32c of libcoper_hip.so when the library is built (False: looked for and absent)


def _native_crc():
    global _native
    if _native is None:
        try:
            from . import _lib
            _native = _lib.load().coper_crc32c
        except Exception:          # no built library (a bare checkout): the NumPy form below is complete on its own
            _native = False
    return _native


def crc32c(data, native=True) -> int:
    """CRC-32C of a bytes-like object: the library's host routine (hardware crc32 instruction, GB/s) when libcoper_hip.so is
    built, else -- or with native=False -- NumPy: large inputs run as parallel lanes and are combined with the zero-shift
    operator (65 MB/s: an 11 MB embedding table checks in well under a second, a 500 MB checkpoint in 8 s)."""
    buf = np.frombuffer(memoryview(data).cast("B"), np.uint8)
    n = buf.size
    if native and n >= 4096 and _native_crc():
        return int(_native_crc()(0, buf.ctypes.data, n))
    lanes = 4096
    if n < 64 * lanes:
        return _crc_update_scalar(0xFFFFFFFF, buf.tobytes()) ^ 0xFFFFFFFF
    m = n // lanes
    body = buf[:m * lanes].reshape(lanes, m)
    st = np.zeros(lanes, np.uint32)
    st[0] = 0xFFFFFFFF                      # only the first lane carries the initial value; the rest start at 0
    for j in range(m):
        st = _T[(st ^ body[:, j]) & 0xFF] ^ (st >> np.uint32(8))
    cols = _zero_shift_matrix(m)
    acc = 0
    for v in st.tolist():
        # acc <- shift(acc, m bytes) xor lane state   (CRC is linear over GF(2) in the register)
        y, i, x = 0, 0, acc
        while x:
            if x & 1:
                y ^= cols[i]
            x >>= 1
            i += 1
        acc = y ^ v
    acc = _crc_update_scalar(acc, buf[m * lanes:].tobytes())
    return acc ^ 0xFFFFFFFF


def masked_crc32c(data) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------------
# varints, protobuf wire format (only what the two bundle messages need)


def _get_varint(b: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        if pos >= len(b):
            raise ValueError("truncated varint")
        c = b[pos]
        pos += 1
        result |= (c & 0x7F) << shift
        if not c & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _put_varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        c = v & 0x7F
        v >>= 7
        if v:
            out.append(c | 0x80)
        else:
            out.append(c)
            return bytes(out)


def _pb_fields(b: bytes) -> Iterable[Tuple[int, int, object]]:
    pos = 0
    while pos < len(b):
        tag, pos = _get_varint(b, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(b, pos)
        elif wt == 1:
            v = b[pos:pos + 8]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(b, pos)
            v = b[pos:pos + n]
            if len(v) != n:
                raise ValueError("truncated protobuf field")
            pos += n
        elif wt == 5:
            v = b[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_shape(b: bytes):
    dims = []
    for f, _, v in _pb_fields(b):
        if f == 2:                                   # Dim
            size = 0
            for ff, _, vv in _pb_fields(v):
                if ff == 1:
                    size = _signed64(vv)
            dims.append(size)
        elif f == 3 and v:
            raise ValueError("tensor of unknown rank in a checkpoint")
    return tuple(dims)


def _parse_entry(b: bytes) -> dict:
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, slices=0)
    for f, _, v in _pb_fields(b):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            e["shape"] = _parse_shape(v)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif f == 7:
            e["slices"] += 1
    return e


def _parse_header(b: bytes) -> dict:
    h = dict(num_shards=0, endianness=0, producer=0)
    for f, _, v in _pb_fields(b):
        if f == 1:
            h["num_shards"] = v
        elif f == 2:
            h["endianness"] = v
        elif f == 3:
            for ff, _, vv in _pb_fields(v):
                if ff == 1:
                    h["producer"] = vv
    return h


def _pb_varint_field(field: int, v: int) -> bytes:
    return _put_varint(field << 3) + _put_varint(v)


def _pb_bytes_field(field: int, b: bytes) -> bytes:
    return _put_varint((field << 3) | 2) + _put_varint(len(b)) + b


# ---------------------------------------------------------------------------------------------------------
# snappy (raw format) decompression: TF writes bundle indexes uncompressed, other table writers may not


def _snappy_uncompress(b: bytes) -> bytes:
    n, pos = _get_varint(b, 0)
    out = bytearray()
    while pos < len(b):
        tag = b[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(b[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += b[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | b[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = b[pos] | (b[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(b[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                        # overlapping copies are legal
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("corrupt snappy block (length)")
    return bytes(out)


# ---------------------------------------------------------------------------------------------------------
# table reader


def _read_block(f: bytes, offset: int, size: int, verify: bool) -> bytes:
    raw = f[offset:offset + size + 5]
    if len(raw) != size + 5:
        raise ValueError("checkpoint index: block handle beyond the end of the file")
    body, ctype = raw[:size], raw[size]
    if verify:
        want = struct.unpack("<I", raw[size + 1:size + 5])[0]
        if masked_crc32c(raw[:size + 1]) != want:
            raise ValueError("checkpoint index: block checksum mismatch")
    if ctype == 0:
        return body
    if ctype == 1:
        return _snappy_uncompress(body)
    raise ValueError("checkpoint index: unknown block compression %d" % ctype)


def _block_entries(block: bytes):
    if len(block) < 4:
        raise ValueError("checkpoint index: block too small")
    nrestart = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * nrestart
    if end < 0:
        raise ValueError("checkpoint index: bad restart array")
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise ValueError("checkpoint index: corrupt block entry")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _table_items(index_bytes: bytes, verify: bool):
    if len(index_bytes) < 48:
        raise ValueError("not a checkpoint index (shorter than a table footer)")
    footer = index_bytes[-48:]
    if struct.unpack("<Q", footer[40:])[0] != TABLE_MAGIC:
        raise ValueError("not a checkpoint index (bad table magic)")
    _, p = _get_varint(footer, 0)        # metaindex offset
    _, p = _get_varint(footer, p)        # metaindex size
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    for _, handle in _block_entries(_read_block(index_bytes, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0)
        bsize, q = _get_varint(handle, q)
        for kv in _block_entries(_read_block(index_bytes, boff, bsize, verify)):
            yield kv


def _shard_name(prefix: str, shard: int, num_shards: int) -> str:
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def list_bundle(prefix: str, verify: bool = True) -> Dict[str, dict]:
    """name -> {dtype (numpy), shape, shard_id, offset, size, crc32c} for every tensor of the checkpoint."""
    with open(prefix + ".index", "rb") as f:
        data = f.read()
    out, header = {}, None
    for k, v in _table_items(data, verify):
        if k == b"":
            header = _parse_header(v)
            continue
        e = _parse_entry(v)
        if e["slices"]:
            raise ValueError("partitioned variable %r: sliced checkpoint entries are not supported" % k.decode())
        if e["dtype"] not in _DTYPES:
            raise ValueError("tensor %r has unsupported dtype id %d" % (k.decode(), e["dtype"]))
        e["dtype"] = _DTYPES[e["dtype"]]
        out[k.decode("utf-8")] = e
    if header is None:
        raise ValueError("checkpoint index has no header entry")
    if header["endianness"] != 0:
        raise ValueError("big-endian checkpoint")
    for e in out.values():
        e["num_shards"] = header["num_shards"]
    return out


def read_bundle(prefix: str, names=None, verify: bool = True) -> Dict[str, np.ndarray]:
    """All (or the named) tensors of `<prefix>.index` + `<prefix>.data-*` as NumPy arrays, checksums verified."""
    entries = list_bundle(prefix, verify)
    if names is not None:
        missing = [n for n in names if n not in entries]
        if missing:
            raise KeyError("not in the checkpoint: %s" % ", ".join(missing))
        entries = {n: entries[n] for n in names}
    out, files = {}, {}
    try:
        for name, e in entries.items():
            fn = _shard_name(prefix, e["shard_id"], e["num_shards"])
            if fn not in files:
                files[fn] = open(fn, "rb")
            f = files[fn]
            f.seek(e["offset"])
            raw = f.read(e["size"])
            count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
            if len(raw) != e["size"] or e["size"] != count * e["dtype"].itemsize:
                raise ValueError("tensor %r: %d bytes on disk, shape %s needs %d" % (name, len(raw), e["shape"],
                                                                                      count * e["dtype"].itemsize))
            if verify and e["crc32c"] is not None and masked_crc32c(raw) != e["crc32c"]:
                raise ValueError("tensor %r: data checksum mismatch" % name)
            out[name] = np.frombuffer(raw, e["dtype"]).reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out


# ---------------------------------------------------------------------------------------------------------
# writer (one shard): lets a model trained here be handed back to `saver.restore` (run_cpg.py:206)


def _block(entries, restart_interval=16) -> bytes:
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(k), len(last))
            while shared < m and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _emit_block(buf: bytearray, body: bytes) -> bytes:
    handle = _put_varint(len(buf)) + _put_varint(len(body))
    buf += body + b"\0" + struct.pack("<I", masked_crc32c(body + b"\0"))
    return handle


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray], block_size: int = 4096) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = []
    header = _pb_varint_field(1, 1) + _pb_varint_field(2, 0) + _pb_bytes_field(3, _pb_varint_field(1, 1))
    items.append((b"", header))
    offset = 0
    with open(_shard_name(prefix, 0, 1), "wb") as fd:
        for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
            a = np.asarray(tensors[name])
            if not a.flags.c_contiguous:
                a = a.copy(order="C")             # (ascontiguousarray would turn a scalar into shape (1,))
            dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
            if np.dtype(dt) not in _DTYPE_IDS:
                raise ValueError("tensor %r: dtype %s has no checkpoint id here" % (name, a.dtype))
            raw = a.astype(dt, copy=False).tobytes()
            shape = b"".join(_pb_bytes_field(2, _pb_varint_field(1, int(s))) for s in a.shape)
            entry = _pb_varint_field(1, _DTYPE_IDS[np.dtype(dt)]) + _pb_bytes_field(2, shape)
            if offset:
                entry += _pb_varint_field(4, offset)
            entry += _pb_varint_field(5, len(raw)) + _put_varint((6 << 3) | 5) + struct.pack("<I", masked_crc32c(raw))
            items.append((name.encode("utf-8"), entry))
            fd.write(raw)
            offset += len(raw)
    buf, index, cur, cur_bytes = bytearray(), [], [], 0
    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 3
        if cur_bytes >= block_size:
            index.append((cur[-1][0], _emit_block(buf, _block(cur))))
            cur, cur_bytes = [], 0
    if cur:
        index.append((cur[-1][0], _emit_block(buf, _block(cur))))
    meta = _emit_block(buf, _block([]))
    idx = _emit_block(buf, _block(index, restart_interval=1))
    footer = meta + idx
    buf += footer + b"\0" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(buf))
