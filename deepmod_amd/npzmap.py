"""Zero-copy reader for the uncompressed `.npz` containers this package writes (feature / packed / raw containers).

`numpy.load` on an `.npz` walks the zip directory per member, copies every member out of the archive and checks its CRC-32
(~0.4 GB/s on the event tables of a raw container: 39 % of a feeder process' time).  The members of an archive written by
`numpy.savez` are STORED, so each array is a contiguous `.npy` image inside the file: map the file once and hand out
`numpy.frombuffer` views.  Compressed or object members fall back to `numpy.load`.  The arrays are read-only.
"""
from __future__ import annotations

import mmap
import re
import struct
import zipfile
from typing import Dict

import numpy as np
from numpy.lib import format as npformat


class _Cursor:
    """Minimal file-like view of a mapped range for numpy's .npy header parser."""

    def __init__(self, mm, pos):
        self.mm, self.pos = mm, pos

    def read(self, n):
        out = self.mm[self.pos:self.pos + n]
        self.pos += len(out)
        return out


ALIGN = 64
_HDR = re.compile(rb"^\{'descr': '([^']+)', 'fortran_order': (True|False), 'shape': \(([0-9, ]*)\), \}\s*$")


def _fast_header(mm, pos):
    """(shape, fortran, dtype, data offset) of the .npy image at `pos`, or None: the header numpy itself writes for a plain dtype is one fixed
    dict literal - parsed with a regular expression instead of ast.literal_eval (20 us per member, eleven members per raw container,
    a thousand containers per feeder: the safe parser was 5 % of a feeder's time).  Anything else (structured dtypes, foreign writers)
    goes to numpy's own parser."""
    if mm[pos:pos + 6] != b'\x93NUMPY':
        return None
    major = mm[pos + 6]
    if major == 1:
        hlen, start = struct.unpack('<H', mm[pos + 8:pos + 10])[0], pos + 10
    elif major in (2, 3):
        hlen, start = struct.unpack('<I', mm[pos + 8:pos + 12])[0], pos + 12
    else:
        return None
    m = _HDR.match(mm[start:start + hlen])
    if m is None:
        return None
    try:
        dtype = np.dtype(m.group(1).decode('latin1'))
    except TypeError:
        return None
    shape = tuple(int(v) for v in m.group(3).replace(b' ', b'').split(b',') if v)
    return shape, m.group(2) == b'True', dtype, start + hlen


def savez_aligned(file, **arrays) -> None:
    """numpy.savez (members STORED), with every member's .npy image starting on a 64-byte boundary of the file: the local zip
    header gets an extra field of padding bytes.  numpy pads the .npy header to a multiple of 64, so the array data of every
    member is then 64-byte aligned in the mapping and `load` hands out aligned views instead of copies.  Readable by numpy.load."""
    import io
    own = isinstance(file, (str, bytes))
    fh = open(file, 'wb') if own else file
    try:
        with zipfile.ZipFile(fh, 'w', compression=zipfile.ZIP_STORED, allowZip64=True) as zf:
            for name, arr in arrays.items():
                buf = io.BytesIO()
                npformat.write_array(buf, np.asanyarray(arr), allow_pickle=False)
                zi = zipfile.ZipInfo(name + '.npy', date_time=(1980, 1, 1, 0, 0, 0))
                zi.compress_type = zipfile.ZIP_STORED
                data = buf.getvalue()
                big = len(data) >= (1 << 31) - 1             # zipfile then adds a 20-byte zip64 extra field in front of ours
                start = fh.tell() + 30 + len(zi.filename.encode()) + (20 if big else 0)
                pad = (-(start + 4)) % ALIGN
                zi.extra = struct.pack('<HH', 0xD3A1, pad) + b'\0' * pad        # private extra-field id, ignored by every reader
                zf.writestr(zi, data)
    finally:
        if own:
            fh.close()


def _central_directory(mm):
    """[(member name, compression method, offset of the local header)] read straight from the mapping (end-of-central-directory record, then the
    fixed 46-byte entries), or None for anything this does not cover (zip64 sizes, an archive comment that hides the record, a damaged file) -
    the caller then asks zipfile.  zipfile's own parser costs 0.2 ms per container in Python objects; a feeder opens a thousand of them."""
    n = len(mm)
    if n < 22:
        return None
    tail = max(0, n - 22 - 1024)
    at = mm.rfind(b'PK\x05\x06', tail)
    if at < 0 or at + 22 > n:
        return None
    n_disk, n_total, cd_size, cd_off = struct.unpack('<HHII', mm[at + 8:at + 20])
    if n_disk != n_total or n_total == 0xFFFF or cd_off == 0xFFFFFFFF or cd_off + cd_size > at:
        return None
    out, pos = [], cd_off
    for _ in range(n_total):
        if pos + 46 > n or mm[pos:pos + 4] != b'PK\x01\x02':
            return None
        method, = struct.unpack('<H', mm[pos + 10:pos + 12])
        csize, usize, n_name, n_extra, n_comment = struct.unpack('<IIHHH', mm[pos + 20:pos + 34])
        ho, = struct.unpack('<I', mm[pos + 42:pos + 46])
        if csize == 0xFFFFFFFF or usize == 0xFFFFFFFF or ho == 0xFFFFFFFF:
            return None
        try:
            name = mm[pos + 46:pos + 46 + n_name].decode('utf-8')
        except UnicodeDecodeError:
            return None
        out.append((name, method, ho))
        pos += 46 + n_name + n_extra + n_comment
    return out


_MADV_POPULATE_READ = 22           # Linux 5.14+: map (and read in) the pages of a range now, like MAP_POPULATE for a whole mapping
_populate_ranges_ok = [hasattr(mmap.mmap, 'madvise')]


def _populate(mm, ranges) -> bool:
    """Map the pages of the byte ranges [(first, end)] of `mm` now.  False: this kernel cannot (the caller maps the whole file instead)."""
    page = mmap.PAGESIZE
    merged = []
    for lo, hi in sorted((lo // page * page, -(-hi // page) * page) for lo, hi in ranges if hi > lo):
        if merged and lo <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], hi)
        else:
            merged.append([lo, hi])
    try:
        for lo, hi in merged:
            mm.madvise(_MADV_POPULATE_READ, lo, min(hi, len(mm)) - lo)
    except (OSError, ValueError):
        return False
    return True


def load(path: str, lazy=()) -> Dict[str, np.ndarray]:
    """name -> array for every member of an .npz file; views into one read-only mapping where the member is stored.
    lazy: members the caller will probably not read (a raw container's basecaller mean / stdv columns: a quarter of its bytes, needed only for a read with
    an empty event) - their pages are neither mapped nor read from the disk until somebody touches them."""
    out: Dict[str, np.ndarray] = {}
    fallback = []
    want_ranges = bool(lazy) and _populate_ranges_ok[0]
    with open(path, 'rb') as fh:
        try:
            # every member is read by whoever loads a container: map the pages in one go instead of one fault per 4 KB (a third of dm_events_merge's
            # time on a 60-byte-per-event table that comes out of the page cache)
            mm = mmap.mmap(fh.fileno(), 0, flags=mmap.MAP_SHARED | (0 if want_ranges else getattr(mmap, 'MAP_POPULATE', 0)), prot=mmap.PROT_READ)
        except (ValueError, OSError):
            mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        infos = _central_directory(mm)
        if infos is None:
            with zipfile.ZipFile(fh) as zf:
                infos = [(i.filename, i.compress_type, i.header_offset) for i in zf.infolist()]
    ranges = []
    for filename, method, ho in infos:
        name = filename[:-4] if filename.endswith('.npy') else filename
        if method != zipfile.ZIP_STORED:
            fallback.append(name)
            continue
        if mm[ho:ho + 4] != b'PK\x03\x04':
            fallback.append(name)
            continue
        n_name, n_extra = struct.unpack('<HH', mm[ho + 26:ho + 30])
        cur = _Cursor(mm, ho + 30 + n_name + n_extra)
        fast = _fast_header(mm, cur.pos)
        if fast is not None:
            shape, fortran, dtype, cur.pos = fast
        else:
            try:
                version = npformat.read_magic(cur)
                shape, fortran, dtype = (npformat.read_array_header_1_0(cur) if version == (1, 0) else npformat.read_array_header_2_0(cur))
            except Exception:
                fallback.append(name)
                continue
        if dtype.hasobject:
            fallback.append(name)
            continue
        count = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        if want_ranges and name not in lazy:
            ranges.append((cur.pos, cur.pos + count * dtype.itemsize))
        if cur.pos % max(dtype.alignment, 1):
            # a member starts wherever the zip local header ends: a view at an offset that is not a multiple of the item alignment
            # would hand misaligned int64 / float64 pointers to the C ABI - such a member is copied
            arr = np.frombuffer(mm, np.uint8, count * dtype.itemsize, cur.pos).copy().view(dtype)
        else:
            arr = np.frombuffer(mm, dtype, count, cur.pos)
        out[name] = arr.reshape(shape, order='F' if fortran else 'C')
    if want_ranges and not _populate(mm, ranges):
        _populate_ranges_ok[0] = False                  # (an older kernel: from now on whole files are mapped at once, as before)
    if fallback:
        z = np.load(path, allow_pickle=False)
        for name in fallback:
            out[name] = z[name]
    return out
