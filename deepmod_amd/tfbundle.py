"""TensorFlow "tensor bundle" (V2 checkpoint) reader and writer, dependency-free.

DeepMod's `detect` restores its BiLSTM from a TF1 checkpoint prefix
(`--modfile`; reference bin/DeepMod_scripts/myDetect.py:955-956 does
`import_meta_graph(prefix+'.meta')` + `restore(latest_checkpoint(dir))`, whose net
effect is "load tensors by variable name").  This module
does exactly that without TensorFlow:

* ``read_index(path)``   parses ``<prefix>.index`` (an SSTable: LevelDB-style
  blocks + 48-byte footer) into ``{name: BundleEntry}``.
* ``load_bundle(prefix)`` returns ``{name: np.ndarray}`` for every float tensor.
* ``write_bundle(prefix, tensors, layout=None)`` writes a valid ``.index`` +
  ``.data-00000-of-00001`` pair (used for synthetic checkpoints in tests/bench,
  because the real ``.data`` shards are absent from the reference tree).

Only what the DeepMod checkpoints use is implemented: one shard, DT_FLOAT,
uncompressed blocks.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
DT_FLOAT = 1


# ----------------------------------------------------------------------------
# varint / protobuf wire helpers
# ----------------------------------------------------------------------------
def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        byte = buf[pos]
        pos += 1
        result |= (byte & 0x7F) << shift
        if not byte & 0x80:
            return result, pos
        shift += 7


def _put_varint(value: int) -> bytes:
    out = bytearray()
    while True:
        byte = value & 0x7F
        value >>= 7
        if value:
            out.append(byte | 0x80)
        else:
            out.append(byte)
            return bytes(out)


def pb_fields(buf: bytes) -> Iterable[Tuple[int, int, object]]:
    """Yield (field_number, wire_type, value) for one protobuf message."""
    pos = 0
    n = len(buf)
    while pos < n:
        tag, pos = _get_varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fno, wt, val


def _pb_tag(fno: int, wt: int) -> bytes:
    return _put_varint((fno << 3) | wt)


# ----------------------------------------------------------------------------
# crc32c (Castagnoli), with TF/LevelDB masking
# ----------------------------------------------------------------------------
def _make_crc_table() -> List[int]:
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)
    return table


_CRC_TABLE = _make_crc_table()
_CRC_TABLE_NP = np.array(_CRC_TABLE, dtype=np.uint32)


def crc32c(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc_mask(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ----------------------------------------------------------------------------
# .index reader
# ----------------------------------------------------------------------------
@dataclass
class BundleEntry:
    name: str
    dtype: int
    shape: Tuple[int, ...]
    shard_id: int
    offset: int
    size: int
    crc32c: Optional[int]


def _read_block(buf: bytes, offset: int, size: int) -> List[Tuple[bytes, bytes]]:
    """Decode one uncompressed SSTable block into [(key, value)]."""
    block = buf[offset:offset + size]
    if offset + size + 1 <= len(buf) and buf[offset + size] != 0:
        raise ValueError("compressed SSTable blocks are not supported")
    num_restarts = struct.unpack_from("<I", block, size - 4)[0]
    limit = size - 4 - 4 * num_restarts
    entries = []
    pos = 0
    key = b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        entries.append((key, block[pos:pos + vlen]))
        pos += vlen
    return entries


def _parse_entry(name: str, value: bytes) -> BundleEntry:
    dtype = 0
    shape: List[int] = []
    shard = offset = size = 0
    crc = None
    for fno, wt, val in pb_fields(value):
        if fno == 1:
            dtype = val
        elif fno == 2:  # TensorShapeProto
            for f2, _, v2 in pb_fields(val):
                if f2 == 2:  # Dim
                    dim = 0
                    for f3, _, v3 in pb_fields(v2):
                        if f3 == 1:
                            dim = v3
                    shape.append(dim)
        elif fno == 3:
            shard = val
        elif fno == 4:
            offset = val
        elif fno == 5:
            size = val
        elif fno == 6:
            crc = struct.unpack("<I", val)[0]
    return BundleEntry(name, dtype, tuple(shape), shard, offset, size, crc)


def read_index(path: str) -> Dict[str, BundleEntry]:
    """Parse ``<prefix>.index`` -> ordered dict of tensor entries (header skipped)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    if len(buf) < 48:
        raise ValueError("%s: too short for an SSTable footer" % path)
    footer = buf[-48:]
    magic = struct.unpack("<Q", footer[40:])[0]
    if magic != TABLE_MAGIC:
        raise ValueError("%s: bad SSTable magic %x" % (path, magic))
    pos = 0
    _meta_off, pos = _get_varint(footer, pos)
    _meta_sz, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_sz, pos = _get_varint(footer, pos)
    entries: Dict[str, BundleEntry] = {}
    for _, handle in _read_block(buf, idx_off, idx_sz):
        boff, hp = _get_varint(handle, 0)
        bsz, hp = _get_varint(handle, hp)
        for key, value in _read_block(buf, boff, bsz):
            if key == b"":
                continue  # BundleHeaderProto
            name = key.decode("utf-8")
            entries[name] = _parse_entry(name, value)
    return entries


def data_path(prefix: str, shard: int = 0, num_shards: int = 1) -> str:
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def load_bundle(prefix: str, names: Optional[Iterable[str]] = None,
                verify_crc: bool = False) -> Dict[str, np.ndarray]:
    """Load float tensors by name from ``<prefix>.index`` / ``.data-00000-of-00001``."""
    entries = read_index(prefix + ".index")
    dpath = data_path(prefix)
    if not os.path.exists(dpath):
        raise FileNotFoundError(
            "%s is missing (only .index/.meta present?) - cannot restore weights" % dpath)
    blob = np.memmap(dpath, dtype=np.uint8, mode="r")
    wanted = list(names) if names is not None else [n for n, e in entries.items() if e.dtype == DT_FLOAT]
    out = {}
    for name in wanted:
        if name not in entries:
            raise KeyError("tensor %r not in checkpoint %s" % (name, prefix))
        e = entries[name]
        if e.dtype != DT_FLOAT:
            raise TypeError("tensor %r is dtype %d, only DT_FLOAT supported" % (name, e.dtype))
        if e.offset + e.size > blob.shape[0]:
            raise ValueError("tensor %r extends past end of %s" % (name, dpath))
        raw = bytes(blob[e.offset:e.offset + e.size])
        if verify_crc and e.crc32c is not None:
            if crc_mask(crc32c(raw)) != e.crc32c:
                raise ValueError("crc32c mismatch for tensor %r" % name)
        out[name] = np.frombuffer(raw, dtype="<f4").reshape(e.shape).copy()
    return out


def latest_checkpoint(directory: str) -> Optional[str]:
    """Counterpart of tf.train.latest_checkpoint: read the `checkpoint` text file."""
    state = os.path.join(directory, "checkpoint")
    if not os.path.exists(state):
        return None
    with open(state) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith("model_checkpoint_path:"):
                val = line.split(":", 1)[1].strip().strip('"')
                return val if os.path.isabs(val) else os.path.join(directory, val)
    return None


# ----------------------------------------------------------------------------
# writer
# ----------------------------------------------------------------------------
def _entry_proto(e: BundleEntry) -> bytes:
    shape = b""
    for d in e.shape:
        dim = _pb_tag(1, 0) + _put_varint(d)
        shape += _pb_tag(2, 2) + _put_varint(len(dim)) + dim
    out = _pb_tag(1, 0) + _put_varint(e.dtype)
    out += _pb_tag(2, 2) + _put_varint(len(shape)) + shape
    if e.shard_id:
        out += _pb_tag(3, 0) + _put_varint(e.shard_id)
    if e.offset:
        out += _pb_tag(4, 0) + _put_varint(e.offset)
    out += _pb_tag(5, 0) + _put_varint(e.size)
    if e.crc32c is not None:
        out += _pb_tag(6, 5) + struct.pack("<I", e.crc32c)
    return out


def _build_block(items: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out = bytearray()
    restarts = []
    prev = b""
    for i, (key, value) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            lim = min(len(prev), len(key))
            while shared < lim and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        out += key[shared:] + value
        prev = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _block_trailer(block: bytes) -> bytes:
    return b"\x00" + struct.pack("<I", crc_mask(crc32c(block + b"\x00")))


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray],
                 layout: Optional[Dict[str, int]] = None,
                 total_size: Optional[int] = None,
                 write_checkpoint_state: bool = True) -> None:
    """Write ``<prefix>.index`` + ``<prefix>.data-00000-of-00001``.

    ``layout`` optionally pins byte offsets per tensor name (to reproduce the real
    DeepMod checkpoint layout with its Adam-slot gaps); otherwise tensors are packed
    back to back in sorted-name order.
    """
    names = sorted(tensors)
    entries: List[BundleEntry] = []
    cursor = 0
    blobs = []
    for name in names:
        arr = np.array(tensors[name], dtype="<f4", order="C")  # (ascontiguousarray would promote 0-d to 1-d)
        raw = arr.tobytes()
        off = layout[name] if layout and name in layout else cursor
        cursor = max(cursor, off + len(raw))
        entries.append(BundleEntry(name, DT_FLOAT, tuple(arr.shape), 0, off, len(raw),
                                   crc_mask(crc32c(raw))))
        blobs.append((off, raw))
    size = max(cursor, total_size or 0)
    data = bytearray(size)
    for off, raw in blobs:
        data[off:off + len(raw)] = raw
    with open(data_path(prefix), "wb") as fh:
        fh.write(bytes(data))

    # BundleHeaderProto{num_shards=1, endianness=LITTLE(0), version{producer=1}}
    header = _pb_tag(1, 0) + _put_varint(1) + _pb_tag(3, 2) + _put_varint(2) + _pb_tag(1, 0) + _put_varint(1)
    items = [(b"", header)] + [(e.name.encode("utf-8"), _entry_proto(e)) for e in entries]
    data_block = _build_block(items)
    out = bytearray()
    data_off = 0
    out += data_block + _block_trailer(data_block)
    meta_block = _build_block([])
    meta_off = len(out)
    out += meta_block + _block_trailer(meta_block)
    last_key = items[-1][0]
    handle = _put_varint(data_off) + _put_varint(len(data_block))
    index_block = _build_block([(last_key + b"\x00", handle)], restart_interval=1)
    idx_off = len(out)
    out += index_block + _block_trailer(index_block)
    footer = _put_varint(meta_off) + _put_varint(len(meta_block)) + _put_varint(idx_off) + _put_varint(len(index_block))
    footer = footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out += footer
    with open(prefix + ".index", "wb") as fh:
        fh.write(bytes(out))
    if write_checkpoint_state:
        d = os.path.dirname(prefix) or "."
        with open(os.path.join(d, "checkpoint"), "w") as fh:
            fh.write('model_checkpoint_path: "%s"\n' % os.path.basename(prefix))
            fh.write('all_model_checkpoint_paths: "%s"\n' % os.path.basename(prefix))
