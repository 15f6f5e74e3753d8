#!/usr/bin/env python3
"""Hand-assembles a TensorFlow V2 checkpoint (`model.ckpt.index` + `model.ckpt.data-00000-of-00001`) BYTE BY BYTE from
the published formats, without importing anything of coper_amd -- so that coper_amd/tf_bundle.py's reader is tested
against bytes its own writer did not produce.  No TensorFlow exists in this environment: this is still a restatement
(by the same author) of tensorflow/core/lib/io/{format,block_builder,table_builder}.cc and
tensorflow/core/protobuf/tensor_bundle.proto, not a TF-written file -- tf_bundle.py stays "format-restated, unpinned".

Layout assembled here (all integers little endian):

  data block  := entry* restart_offset(fixed32)* num_restarts(fixed32)
  entry       := varint32 shared_key_bytes | varint32 unshared_key_bytes | varint32 value_bytes | key suffix | value
  on disk     := block | type byte (0 = uncompressed) | fixed32 mask(crc32c(block | type byte))
  mask(c)     := ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32
  index block := one entry per data block: key >= last key of the block (< first key of the next), value = BlockHandle
  BlockHandle := varint64 offset | varint64 size            (size without the 5-byte trailer)
  footer      := metaindex BlockHandle | index BlockHandle | zero padding to 40 bytes | fixed64 0xdb4775248b80fb57

  key ""  -> BundleHeaderProto : 08 01 (num_shards = 1)  [endianness LITTLE = 0: proto3 omits it]  1a 02 08 01 (version{producer: 1})
  key k   -> BundleEntryProto  : 08 <dtype>  12 <len> <TensorShapeProto>  [18 shard_id = 0 omitted]  [20 <offset> omitted when 0]
                                 28 <size>  35 <fixed32 masked crc32c of the tensor bytes>
  TensorShapeProto: (12 02 08 <dim>)* ; a scalar has an empty one (12 00)

Run:  python tests/golden/make_tf_bundle_handmade.py   -> tests/golden/tf_bundle_handmade.json (hex strings + expected values)."""
import json
import os
import struct


def crc32c(data: bytes) -> int:
    """Bitwise CRC-32C (Castagnoli), reflected polynomial 0x82F63B78 -- deliberately the slow textbook form."""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


assert crc32c(b"123456789") == 0xE3069283          # the check value of the CRC catalogue / RFC 3720 appendix B.4
assert crc32c(bytes(32)) == 0x8A9136AA             # RFC 3720 B.4: 32 bytes of zeros


def mask(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def entry(shared, key_suffix, value):
    return varint(shared) + varint(len(key_suffix)) + varint(len(value)) + key_suffix + value


def block(entries_bytes, restart_offsets):
    b = entries_bytes
    for r in restart_offsets:
        b += struct.pack("<I", r)
    return b + struct.pack("<I", len(restart_offsets))


def on_disk(blk):
    return blk + b"\x00" + struct.pack("<I", mask(crc32c(blk + b"\x00")))


f32 = lambda *v: struct.pack("<%df" % len(v), *v)
ent_emb = f32(0.5, -1.25, 2.0, 3.5, -0.75, 8.0)     # [3, 2]
pred_bias = f32(0.125, -0.5, 1.5)                    # [3]
rel_emb = f32(1.0, 2.0, -3.0, 4.0)                   # [2, 2]
global_step = struct.pack("<q", 7)                   # int64 scalar

# data file: tensors in key order
names = ["global_step", "variables/variables/ent_emb", "variables/variables/pred_bias", "variables/variables/rel_emb"]
payload = {"global_step": global_step, "variables/variables/ent_emb": ent_emb, "variables/variables/pred_bias": pred_bias,
           "variables/variables/rel_emb": rel_emb}
data, offsets = b"", {}
for n in names:
    offsets[n] = len(data)
    data += payload[n]


def dims(*d):
    return b"".join(b"\x12\x02\x08" + bytes([x]) for x in d)


def bundle_entry(dtype, shape_proto, offset, size, tensor_bytes):
    e = b"\x08" + bytes([dtype]) + b"\x12" + bytes([len(shape_proto)]) + shape_proto
    if offset:
        e += b"\x20" + varint(offset)
    return e + b"\x28" + varint(size) + b"\x35" + struct.pack("<I", mask(crc32c(tensor_bytes)))


DT_FLOAT, DT_INT64 = 1, 9
header = b"\x08\x01" + b"\x1a\x02\x08\x01"
e_step = bundle_entry(DT_INT64, b"", offsets["global_step"], 8, global_step)
e_ent = bundle_entry(DT_FLOAT, dims(3, 2), offsets["variables/variables/ent_emb"], 24, ent_emb)
e_bias = bundle_entry(DT_FLOAT, dims(3), offsets["variables/variables/pred_bias"], 12, pred_bias)
e_rel = bundle_entry(DT_FLOAT, dims(2, 2), offsets["variables/variables/rel_emb"], 16, rel_emb)

# data block 1: "", "global_step", "variables/variables/ent_emb" -- restart interval 2: a second restart point at entry 3
b1_e1 = entry(0, b"", header)
b1_e2 = entry(0, b"global_step", e_step)
b1_e3 = entry(0, b"variables/variables/ent_emb", e_ent)                # a restart point: shared = 0
blk1 = block(b1_e1 + b1_e2 + b1_e3, [0, len(b1_e1) + len(b1_e2)])
# data block 2: "variables/variables/pred_bias", then "variables/variables/rel_emb" sharing the 20-byte prefix "variables/variables/"
b2_e1 = entry(0, b"variables/variables/pred_bias", e_bias)
b2_e2 = entry(20, b"rel_emb", e_rel)
blk2 = block(b2_e1 + b2_e2, [0])

index_file = b""
h1 = (len(index_file), len(blk1)); index_file += on_disk(blk1)
h2 = (len(index_file), len(blk2)); index_file += on_disk(blk2)
meta = block(b"", [0])                                                  # empty metaindex block
hm = (len(index_file), len(meta)); index_file += on_disk(meta)
handle = lambda h: varint(h[0]) + varint(h[1])
# index block: separator "variables/variables/f" (> ...ent_emb, < ...pred_bias), and a short successor "w" of the last key
idx = block(entry(0, b"variables/variables/f", handle(h1)) + entry(0, b"w", handle(h2)), [0])
hi = (len(index_file), len(idx)); index_file += on_disk(idx)
footer = handle(hm) + handle(hi)
footer += bytes(40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
index_file += footer
assert len(footer) == 48

out = {"_doc": "hand-assembled TF V2 checkpoint (tests/golden/make_tf_bundle_handmade.py); hex of the two files + the values they hold",
       "index_hex": index_file.hex(), "data_hex": data.hex(),
       "tensors": {"global_step": {"dtype": "int64", "shape": [], "values": [7]},
                   "variables/variables/ent_emb": {"dtype": "float32", "shape": [3, 2], "values": [0.5, -1.25, 2.0, 3.5, -0.75, 8.0]},
                   "variables/variables/pred_bias": {"dtype": "float32", "shape": [3], "values": [0.125, -0.5, 1.5]},
                   "variables/variables/rel_emb": {"dtype": "float32", "shape": [2, 2], "values": [1.0, 2.0, -3.0, 4.0]}}}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf_bundle_handmade.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, len(index_file), "index bytes,", len(data), "data bytes")
