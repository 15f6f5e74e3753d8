#!/usr/bin/env python3
"""Hand-assembles a TFRecord file of `tf.train.Example` samples BYTE BY BYTE from the published formats, without importing
anything of coper_amd -- so that coper_amd/tf_records.py's reader is tested against bytes its own writer did not produce.
No TensorFlow exists in this environment: this is a restatement (by the same author) of
tensorflow/core/lib/io/record_writer.cc and tensorflow/core/example/{example,feature}.proto, not a TF-written file.

  record   := fixed64 length | fixed32 mask(crc32c(length bytes)) | data | fixed32 mask(crc32c(data))        (little endian)
  mask(c)  := ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32
  Example  := 0a <len> Features
  Features := (0a <len> MapEntry)*                       map<string, Feature> feature = 1
  MapEntry := 0a <len> key | 12 <len> Feature
  Feature  := 1a <len> Int64List                          (oneof kind: bytes_list = 1, float_list = 2, int64_list = 3)
  Int64List:= 0a <len> varint*                            (repeated int64 value = 1 [packed = true]; negative values
                                                           are 10-byte two's-complement varints)
             or (08 varint)*                               (the unpacked form: legal on the wire, parsers accept both)

The schema is the reference loader's (`data.py:574-594`): int64 features e1, e2, rel, e2_multi (variable length) and
is_inverse.  Map entries are emitted in the sorted key order protobuf's deterministic serialisation uses for maps; one
sample uses a shuffled order and the unpacked list form.

Run:  python tests/golden/make_tfrecord_handmade.py   -> tests/golden/tfrecord_handmade.json (hex string + expected values)."""
import json
import os
import struct


def crc32c(data: bytes) -> int:
    """Bitwise CRC-32C (Castagnoli), reflected polynomial 0x82F63B78 -- the slow textbook form."""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


assert crc32c(b"123456789") == 0xE3069283          # the check value of the CRC catalogue / RFC 3720 appendix B.4


def mask(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(v):
    v &= (1 << 64) - 1                             # int64 -> two's complement
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def ld(field, payload):                            # length-delimited field
    return bytes([(field << 3) | 2]) + varint(len(payload)) + payload


def int64_list(values, packed=True):
    if packed:
        return ld(1, b"".join(varint(v) for v in values)) if values else b""
    return b"".join(bytes([(1 << 3) | 0]) + varint(v) for v in values)


def example(sample, order=None, packed=True):
    keys = order or sorted(sample)
    feats = b"".join(ld(1, ld(1, k.encode()) + ld(2, ld(3, int64_list(sample[k], packed)))) for k in keys)
    return ld(1, feats)


def record(data):
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", mask(crc32c(head))) + data + struct.pack("<I", mask(crc32c(data)))


SAMPLES = [
    dict(e1=[3], e2=[17], rel=[2], e2_multi=[17, 4, 129], is_inverse=[0]),
    dict(e1=[0], e2=[0], rel=[0], e2_multi=[0], is_inverse=[1]),
    # ids that need 2-, 3- and 5-byte varints; a long filter list (length field of the packed list > 127 bytes)
    dict(e1=[14540], e2=[300], rel=[473], e2_multi=[(i * 7919) % 9999991 for i in range(70)], is_inverse=[0]),
    # an empty filter list (an Int64List with no values), a negative value (10-byte varint)
    dict(e1=[5], e2=[6], rel=[7], e2_multi=[], is_inverse=[-1]),
]


def main():
    blob = b""
    blob += record(example(SAMPLES[0]))
    blob += record(example(SAMPLES[1], order=["rel", "is_inverse", "e2_multi", "e1", "e2"], packed=False))
    blob += record(example(SAMPLES[2]))
    blob += record(example(SAMPLES[3]))
    # known answers of the framing, spelled out for the first record
    first = example(SAMPLES[0])
    head = struct.pack("<Q", len(first))
    out = dict(file_hex=blob.hex(), samples=SAMPLES, first_length=len(first), first_length_crc=mask(crc32c(head)),
               first_data_crc=mask(crc32c(first)))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tfrecord_handmade.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, len(blob), "bytes")


if __name__ == "__main__":
    main()
