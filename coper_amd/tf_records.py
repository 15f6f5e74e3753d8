"""Reader / writer for the TFRecord files the reference's loader keeps next to its JSON files
(`data.py:353-397`: `<split>-<n>.tfrecords`, one `tf.train.Example` per sample with int64 features `e1`, `e2`, `rel`,
`e2_multi` (variable length) and `is_inverse`, `data.py:574-594`), SURVEY.md 8f-2.  No TensorFlow: the framing and the
Example protobuf are restated from their published layout (tensorflow/core/lib/io/record_writer.cc,
tensorflow/core/example/{example,feature}.proto); round trip + known framing values are what the tests pin.

  record   uint64 length | uint32 masked_crc32c(length bytes) | data[length] | uint32 masked_crc32c(data)   (little endian)
  Example  {1: Features {1: map<string, Feature> entry {1: key, 2: Feature {3: Int64List {1: packed or repeated varint}}}}}
"""
from __future__ import annotations

import glob
import os
import struct
from typing import Dict, Iterable, Iterator, List

import numpy as np

from .tf_bundle import _get_varint, _pb_bytes_field, _pb_fields, _put_varint, _signed64, masked_crc32c

__all__ = ["read_records", "write_records", "parse_example", "encode_example", "read_split", "write_split"]


def read_records(path, verify=True) -> Iterator[bytes]:
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise ValueError("%s: truncated record header" % path)
            (n,), (c,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if verify and masked_crc32c(head[:8]) != c:
                raise ValueError("%s: corrupt record length" % path)
            body = f.read(n + 4)
            if len(body) != n + 4:
                raise ValueError("%s: truncated record" % path)
            if verify and masked_crc32c(body[:n]) != struct.unpack("<I", body[n:])[0]:
                raise ValueError("%s: corrupt record data" % path)
            yield body[:n]


def write_records(path, records: Iterable[bytes]) -> int:
    n = 0
    with open(path, "wb") as f:
        for r in records:
            head = struct.pack("<Q", len(r))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + r + struct.pack("<I", masked_crc32c(r)))
            n += 1
    return n


def _int64_list(b: bytes) -> List[int]:
    out = []
    for f, wt, v in _pb_fields(b):
        if f != 1:
            continue
        if wt == 2:                       # packed (what the TF writers emit)
            pos = 0
            while pos < len(v):
                x, pos = _get_varint(v, pos)
                out.append(_signed64(x))
        else:                             # unpacked repeated varint (legal on the wire)
            out.append(_signed64(v))
    return out


def parse_example(b: bytes) -> Dict[str, List[int]]:
    """tf.train.Example bytes -> {feature name: int64 values} (float / bytes features are ignored: the schema of
    data.py:574-594 has none)."""
    out = {}
    for f, _, feats in _pb_fields(b):
        if f != 1:
            continue
        for f2, _, entry in _pb_fields(feats):
            if f2 != 1:
                continue
            key, val = None, None
            for f3, _, v in _pb_fields(entry):
                if f3 == 1:
                    key = v.decode("utf-8")
                elif f3 == 2:
                    val = v
            if key is None or val is None:
                continue
            for f4, _, v in _pb_fields(val):
                if f4 == 3:               # Int64List
                    out[key] = _int64_list(v)
    return out


def encode_example(features: Dict[str, Iterable[int]]) -> bytes:
    entries = b""
    for key in sorted(features):          # map entries in key order, as the TF serializer's deterministic mode
        packed = b"".join(_put_varint(int(x)) for x in features[key])
        int64_list = _pb_bytes_field(1, packed) if packed else b""
        feature = _pb_bytes_field(3, int64_list)
        entries += _pb_bytes_field(1, _pb_bytes_field(1, key.encode("utf-8")) + _pb_bytes_field(2, feature))
    return _pb_bytes_field(1, entries)


def read_split(directory, filetype, include_inv_relations=False, verify=True):
    """`<directory>/<filetype>-*.tfrecords` (data.py:360) -> id arrays + CSR filter (the form `EvalDataset` and
    `TrainDataset` take), inverse-relation samples dropped unless asked for (data.py:113-114,139-140)."""
    files = sorted(glob.glob(os.path.join(directory, "%s-*.tfrecords" % filetype)),
                   key=lambda p: int(os.path.basename(p).split("-")[-1].split(".")[0]))
    if not files:
        raise FileNotFoundError("no %s-*.tfrecords under %s" % (filetype, directory))
    e1, e2, rel, indptr, idx = [], [], [], [0], []
    for path in files:
        for rec in read_records(path, verify):
            ex = parse_example(rec)
            if ex.get("is_inverse", [0])[0] and not include_inv_relations:
                continue
            e1.append(ex["e1"][0]); e2.append(ex["e2"][0]); rel.append(ex["rel"][0])
            idx.extend(sorted(set(ex.get("e2_multi", []))))
            indptr.append(len(idx))
    return dict(e1=np.asarray(e1, np.int64), e2=np.asarray(e2, np.int64), rel=np.asarray(rel, np.int64),
                filt_indptr=np.asarray(indptr, np.int64), filt_idx=np.asarray(idx, np.int64))


def write_split(directory, filetype, samples, max_records_per_file=1000000):
    """samples: iterable of dicts with int e1, e2, rel, list e2_multi, bool is_inverse -> the reference's files
    (`<filetype>-<n>.tfrecords`, at most max_records_per_file each, data.py:341,364-385).  Returns the file names."""
    os.makedirs(directory, exist_ok=True)
    names, buf, index = [], [], 0

    def flush():
        nonlocal buf, index
        path = os.path.join(directory, "%s-%d.tfrecords" % (filetype, index))
        write_records(path, buf)
        names.append(path)
        buf, index = [], index + 1

    for s in samples:
        buf.append(encode_example(dict(e1=[s["e1"]], e2=[s["e2"]], rel=[s["rel"]], e2_multi=list(s["e2_multi"]),
                                       is_inverse=[1 if s.get("is_inverse", False) else 0])))
        if len(buf) >= max_records_per_file:
            flush()
    if buf or not names:
        flush()
    return names
