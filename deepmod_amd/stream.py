"""Streaming, read-sharded detect: one process per GPU keeps the per-position counters on the device for the
whole run and the only cross-process exchange is one integer RCCL reduce per contig x strand at the end.

The reference moves every read's predictions through the file system: detect workers write per-read tables
(myDetect.py:716-760) and per-chromosome index files (:762-782, :1194-1221), summary workers read them back and
accumulate a dict keyed by (chr, strand, pos) (:1028-1120); several runs are merged by adding BED files
(DeepMod_tools/sum_chr_mod.py:47-52).  Here the same arithmetic is

    worker batch -> feature rows + per-row (position, flags)           host, feeder threads   (`prepare_*`)
                 -> dm_predict_read (windows assembled on the device)   one in-order device queue per GPU
                 -> dm_summary_add_classified / dm_summary_add          (class -> base scatter of :824-833 fused with :1089-1100)
    end of run   -> dm_summary_reduce over a persistent communicator    deepmod_amd/comm.py
                 -> rank 0: dm_summary_fetch -> BED bytes               deepmod_amd/summary.py:bed_lines

and the BED files are byte-identical to the stored path's (tests/test_gpu_stream.py) for any number of ranks:
integer sums do not depend on the order or the sharding of the reads.

`StreamEngine` is the rank-local logic; it talks to the device through a small backend object (`HipBackend`: the C ABI).
tests/test_stream_gloo.py drives the same engine on CPU ranks with a stand-in backend (oracle classifier and counters,
gloo transport) - only the transport and the device calls are replaced, the sharding / grouping / merge logic is this file.
"""
from __future__ import annotations

import ctypes
import json
import os
import queue
import threading
import time
from collections import defaultdict
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from . import predstore, rawreads

PAD = 100          # zero-padded feature rows on both sides of a read (myDetect.py:850-851)
HALF = 10          # windowsize // 2


class Prepared:
    """One worker batch, ready for the device.  Rows of the reads are concatenated, reads grouped by (contig, strand)."""
    __slots__ = ('_rows', 'pos', 'flags', 'n_rows', 'groups', 'n_windows', 'n_reads', 'errors', 'contig_len', 'timing', 'files', 'on_done', 'f32', 'sel',
                 'ev3', 'code', 'rdesc', 'sig')

    @property
    def rows(self):
        """Feature rows [n_rows][7].  A batch in the device form (ev3 / code / rdesc, see below) has none on the host: a consumer that asks
        for them anyway (the CPU backend of the tests, a comparison) gets the host restatement of dm_rows_assemble."""
        if self._rows is None and self.ev3 is not None:
            return assemble_rows(self.ev3, self.code, self.rdesc, self.n_rows)
        if self._rows is None and self.sig is not None:
            raise ValueError('the event statistics of this batch stay on the device (resident form): its feature rows exist only there')
        return self._rows

    @rows.setter
    def rows(self, value):
        self._rows = value

    def __init__(self):
        self._rows = np.zeros((0, 7), np.float32)
        # device form of a batch of raw reads (round 5, rowsbatch.inc dm_rows_emit_device): ev3 f32[E][3] = (mean, stdv, length) of the events the rows
        # show, code u8[n_rows] = one-hot class of a row (255: none), rdesc i64[reads][4] = (first row, row -> event shift, first event, end event);
        # `rows` is then built on the device (HipBackend.submit -> dm_rows_assemble): 13 instead of 28 bytes per row cross the host and PCIe
        self.ev3 = self.code = self.rdesc = None
        # resident form (round 6): sig = (feeder, request number) of the signal request whose statistics block - (mean, stdv, length) of every merged event
        # of the batch, written by dm_signal_event_stats_device - stays on the device; code / rdesc as above, rdesc's event indices point into that block
        self.sig = None
        self.pos = np.zeros(0, np.int64)          # [n_rows classified rows | extra rows]
        self.flags = np.zeros(0, np.uint8)
        self.n_rows = 0
        self.groups: List[Tuple] = []   # (chr, strand, row_lo, row_hi, extra_lo, extra_hi[, sel_lo, sel_hi])
        # compact form (the compiled path's default): sel = int32 feature-row indices of the windows centred on a base of interest -
        # the only ones whose class can reach the BED (myDetect.py:1091); pos / flags then hold [len(sel) | extras] entries and
        # groups carry (sel_lo, sel_hi).  None: classic form, pos / flags hold one entry per feature row and every window is classified.
        self.sel = None
        self.n_windows = 0
        self.n_reads = 0
        self.errors: Dict[str, List[str]] = defaultdict(list)
        self.contig_len: Dict[str, int] = {}
        self.timing: Dict[str, float] = defaultdict(float)
        self.files: List[str] = []
        self.f32 = False             # a feature outside the split-f16 kernel's range (or NaN): this batch runs the fp32 kernel
        self.on_done = None          # called once the device has consumed the host arrays (a feeder slot goes back to its queue)


def assemble_rows(ev3: np.ndarray, code: np.ndarray, rdesc: np.ndarray, n_rows: int) -> np.ndarray:
    """Host restatement of dm_rows_assemble (deepmod_hip.hip rows_assemble_kernel): the device form of a raw batch -> rows [n_rows][7] =
    one-hot of the row's reference base | mean, stdv, length of the event it shows (get_Feature, myDetect.py:839-903)."""
    rows = np.zeros((n_rows, 7), np.float32)
    if n_rows == 0:
        return rows
    code = np.asarray(code[:n_rows])
    for c in range(4):
        rows[code == c, c] = 1.0
    rdesc = np.asarray(rdesc, np.int64).reshape(-1, 4)
    q = np.arange(n_rows, dtype=np.int64)
    r = np.searchsorted(rdesc[:, 0], q, side='right') - 1          # the last read whose first row is <= q
    e = q + rdesc[r, 1]
    has = (e >= rdesc[r, 2]) & (e < rdesc[r, 3])
    rows[has, 4:7] = np.asarray(ev3, np.float32).reshape(-1, 3)[e[has]]
    return rows


def rows_from_packed(pk: Dict, base: str, src: str, out: Prepared) -> None:
    """Packed container arrays -> device-ready rows, appended to `out` as per-read pieces (see `finish`).
    Vectorised over all reads of the container.  Per read (arguments of the reference's mPredict1, myDetect.py:787-834):
    the k-th aligned event (k < n = events - clips) is the k-th table row whose readbase is not '-' and its window is
    centred on feature row 100 + k; sum_handler (:1089-1100) then counts, for table rows with refbase == Base:
    touch, cov if readbase != '-', mod if that row was classified 1."""
    ro, bo, eo = pk['row_off'], pk['bmi_off'], pk['ev_off']
    nreads = len(pk['reads'])
    if nreads == 0:
        return
    meta = pk['reads']
    start_clip = np.array([m['start_clip'] for m in meta], np.int64)
    end_clip = np.array([m['end_clip'] for m in meta], np.int64)
    # clips come from a container's metadata: classified BEFORE any arithmetic with them (values near the ends of int64 would wrap the
    # subtraction into a plausible count) - negative: an index error of the ledger; too large for the read: no aligned event = "Less Event"
    # (the same rule as rowsbatch.inc plan_read: the ledger keys of the two build paths agree)
    nev = eo[1:] - eo[:-1]
    clip_neg = (start_clip < 0) | (end_clip < 0)
    clip_over = ~clip_neg & ((start_clip > nev) | (end_clip > nev - np.minimum(start_clip, nev)))
    start_clip = np.where(clip_neg | clip_over, 0, start_clip)
    end_clip = np.where(clip_neg | clip_over, 0, end_clip)
    n_al = np.where(clip_neg | clip_over, 0, nev - start_clip - end_clip)     # aligned events per read
    readb, refb, refi = pk['readbase'], pk['refbase'], pk['refbasei']
    not_gap = readb != b'-'
    is_base = refb == base.encode('ascii')                                # Base is one of ACGT: never '-', 'N', 'n'
    cnt = np.cumsum(not_gap, dtype=np.int64)
    excl = np.append(cnt - not_gap, cnt[-1] if len(cnt) else 0)          # aligned rows before row j (whole container); [B] = total
    read_of = np.repeat(np.arange(nreads), bo[1:] - bo[:-1])
    first_cnt = excl[bo[:-1]]
    k = excl[:-1] - first_cnt[read_of]                                     # rank of an aligned row within its read
    n_have = excl[bo[1:]] - first_cnt                                      # aligned table rows per read
    ok = (n_al >= 50) & (n_have >= n_al) & ((ro[1:] - ro[:-1]) == n_al + 2 * PAD)
    for i in np.flatnonzero(~ok):
        if clip_neg[i]:
            out.errors["Prediction failed: IndexError"].append(src)
        elif n_al[i] < 50:
            out.errors["Less Event"].append(src)                          # myDetect.py:702-705
        else:
            out.errors["Prediction failed: IndexError"].append(src)       # fewer aligned table rows than aligned events
    # event base vs table base of every aligned row (the reference prints and goes on, :826-828)
    ev_base = pk['evbase']
    sel = not_gap & (k < n_al[read_of]) & ok[read_of]
    ev_idx = eo[:-1][read_of] + start_clip[read_of] + k
    bad = np.flatnonzero(sel & (ev_base[np.minimum(ev_idx, max(len(ev_base) - 1, 0))] != readb)) if len(ev_base) else []
    for j in bad[:20]:
        print('Error Does not match', readb[j].decode(), ev_base[ev_idx[j]].decode(), int(j - bo[read_of[j]]), int(k[j] + start_clip[read_of[j]]))
    nrows = len(pk['tx'])
    pos_row = np.zeros(nrows, np.int64)
    flag_row = np.zeros(nrows, np.uint8)
    dst = ro[:-1][read_of] + PAD + k
    pos_row[dst[sel]] = refi[sel]
    flag_row[dst[sel]] = is_base[sel].astype(np.uint8) | np.uint8(2)
    extra = is_base & ~sel & ok[read_of]                                   # deletion rows (and aligned rows past n): no window
    ex_idx = np.flatnonzero(extra)
    ex_read = read_of[ex_idx]
    ex_flag = np.uint8(1) | (not_gap[ex_idx].astype(np.uint8) << 1)
    ex_off = np.searchsorted(ex_read, np.arange(nreads + 1))
    for i in np.flatnonzero(ok):
        m = meta[i]
        out._pieces.append((m['chr'], m['strand'], pk['tx'][ro[i]:ro[i + 1]], pos_row[ro[i]:ro[i + 1]], flag_row[ro[i]:ro[i + 1]],
                            refi[ex_idx[ex_off[i]:ex_off[i + 1]]], ex_flag[ex_off[i]:ex_off[i + 1]], int(n_al[i])))
    for c, ln in pk.get('contig_len', {}).items():
        out.contig_len[c] = max(out.contig_len.get(c, 0), int(ln))


def rows_from_reads(reads: Iterable[Dict], base: str, src: str, out: Prepared) -> None:
    """Classic read dicts (mfeatures, base_map_info, events, clips: what readmap.map_records and the format-1 feature
    containers produce) -> the same pieces, through the packed layout."""
    reads = list(reads)
    if not reads:
        return
    tx, refb, readb, refi, evb, metas = [], [], [], [], [], []
    for rd in reads:
        bmi = rd['base_map_info']
        tx.append(np.asarray(rd['mfeatures'][:, 3:], np.float32))
        if 'table_s1' in rd:                       # reads that came through dm_map_read carry the byte columns already
            rb, qb, ri = rd['table_s1']
            refb.append(rb); readb.append(qb); refi.append(ri)
        else:
            refb.append(predstore.u1_to_s1(bmi['refbase']))
            readb.append(predstore.u1_to_s1(bmi['readbase']))
            refi.append(bmi['refbasei'].astype(np.int64))
        evb.append(predstore.u1_to_s1(rawreads.event_bases(rd['events']['model_state'])))
        metas.append(rd)
    off = lambda parts: np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    pk = {'tx': np.concatenate(tx), 'refbase': np.concatenate(refb), 'readbase': np.concatenate(readb),
          'refbasei': np.concatenate(refi), 'evbase': np.concatenate(evb), 'row_off': off(tx), 'bmi_off': off(refb),
          'ev_off': off(evb), 'reads': metas, 'contig_len': {}}
    rows_from_packed(pk, base, src, out)


def finish(out: Prepared, alloc=None) -> Prepared:
    """Group the collected reads by (contig, strand) and build the contiguous host arrays of the batch.
    alloc(n_rows, n_pos) -> (rows[n_rows, 7] f32, pos[n_pos] i64, flags[n_pos] u8) lets a feeder process build them directly in
    shared memory."""
    pieces = out._pieces
    order = sorted(range(len(pieces)), key=lambda i: (pieces[i][0], pieces[i][1]))
    rows, pos, flags, xpos, xflags = [], [], [], [], []
    r = x = 0
    cur = None
    for i in order:
        c, s, tx, p, f, xp, xf, n = pieces[i]
        if cur is None or (c, s) != cur[:2]:
            if cur is not None:
                out.groups.append((cur[0], cur[1], cur[2], r, cur[3], x))
            cur = (c, s, r, x)
        rows.append(tx); pos.append(p); flags.append(f); xpos.append(xp); xflags.append(xf)
        r += len(tx)
        x += len(xp)
        out.n_windows += n
        out.n_reads += 1
        if len(p):
            mx = int(max(p.max(), xp.max() if len(xp) else 0)) + 1
            if mx > out.contig_len.get(c, 0):
                out.contig_len[c] = mx            # lower bound when no reference length is known
    if cur is not None:
        out.groups.append((cur[0], cur[1], cur[2], r, cur[3], x))
    if rows and alloc is not None:
        out.rows, out.pos, out.flags = alloc(r, r + x)
        np.concatenate(rows, out=out.rows, casting='same_kind')
        np.concatenate(pos + xpos, out=out.pos, casting='same_kind')
        np.concatenate(flags + xflags, out=out.flags, casting='same_kind')
    elif rows:
        out.rows = np.ascontiguousarray(np.concatenate(rows), np.float32)
        out.pos = np.concatenate(pos + xpos)
        out.flags = np.concatenate(flags + xflags)
    out.n_rows = r
    out._pieces = []
    out.f32 = not in_f16_range(out.rows)
    return out


def in_f16_range(rows: np.ndarray) -> bool:
    """True if the split-f16 kernel takes these feature rows (include/deepmod_hip.h: |x| <= 65504; an event length may go up
    to 65504 * 2^k, but a read with an event that long is rare enough to take the fp32 kernel with the rest of its batch).
    NaN compares false: not in range."""
    if rows.size == 0:
        return True
    return bool(rows.max() <= 65504.0) and bool(rows.min() >= -65504.0)


class _PreparedBuilder(Prepared):
    __slots__ = ('_pieces',)

    def __init__(self):
        super().__init__()
        self._pieces = []


_REF_BYTES: Dict[Optional[str], Dict[str, bytes]] = {}      # reference path -> contig -> ASCII bytes (encoded once per feeder process)
_ROWS_ERRORS = {1: "Less Event", 2: "Prediction failed: IndexError", 4: "CIGAR-Error", 6: "No reference sequence", 7: "Error Does not match"}


def _c_arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _prepare_batch_c(moptions, files: List[str], make_normalizer=None, alloc=None) -> Prepared:
    """prepare_batch with the per-read work behind the C ABI: one dm_events_merge per raw container, one signal request, one
    dm_rows_add_raw / dm_rows_add_packed per input kind, dm_rows_info + dm_rows_emit straight into the hand-over arrays.
    Same Prepared (rows, pos, flags, groups, errors) as the Python path below, which stays as the restatement the tests compare
    with (tests/test_stream_feeders.py)."""
    import ctypes
    from . import _lib, detect, readmap
    lib = _lib.load()
    out = Prepared()
    out.files = list(files)
    base = moptions['Base']
    h = lib.dm_rows_create(base.encode('ascii'))
    if not h:
        raise _lib.DeepModHipError("dm_rows_create: " + _lib.last_error())
    keep = []                       # arrays the handle borrows until emit
    contigs: Dict[str, int] = {}
    srcs: List[str] = []            # source file of every read, in the order the reads were added
    strands_c = {'+': 0, '-': 1}
    try:
        t0 = time.perf_counter()
        raw_files = [f for f in files if f.endswith(rawreads.RAW_SUFFIX)]
        if raw_files:
            from . import npzmap
            normalizer = make_normalizer() if make_normalizer else None
            if normalizer is None:
                from . import signal as dmsignal
                normalizer = dmsignal.SignalNormalizer(int(moptions.get('device', 0)))
            ids, id_src, raw_parts, raw_offs, ev_offs = [], [], [], [0], [0]
            merges = []                     # per container that was merged: the arguments of its dm_events_merge call (kept alive), for _fallback_values
            opened = []
            for f5f in raw_files:
                try:
                    z = npzmap.load(f5f, lazy=('ev_mean', 'ev_stdv'))
                    if 'format' not in z:
                        raise ValueError('format-1 raw container')
                    meta = json.loads(str(z['meta']))
                    n = len(meta)
                    eo = _c_arr(z['ev_off'], np.int64)
                    if len(eo) != n + 1:
                        raise ValueError('event offsets of a damaged container')
                    ms = _c_arr(z['ev_model_state'], z['ev_model_state'].dtype)
                    args = [_c_arr(z['ev_mean'], np.float64), _c_arr(z['ev_stdv'], np.float64), _c_arr(z['ev_start'], np.uint64),
                            _c_arr(z['ev_length'], np.uint64)]
                    mv = _c_arr(z['ev_move'], np.int64)
                    ro = _c_arr(z['raw_off'], np.int64)
                    if len(ro) != n + 1 or ro[0] != 0 or (np.diff(ro) < 0).any() or ro[-1] > len(z['raw']):
                        raise ValueError('signal offsets of a damaged container')
                    opened.append((f5f, z, meta, n, eo, max(int(eo[-1]), 0), ms, args, mv, ro))
                except Exception:
                    out.errors["Cannot open fast5 or other errors"].append(f5f)
                    print("Cannot open fast5 or other errors: {}".format(f5f))
            # the merged event tables of the whole batch, written in place container after container (no per-container pieces to concatenate)
            cap_ev = sum(o[5] for o in opened)
            m_start, m_len, m_base = np.empty(max(cap_ev, 1), np.uint64), np.empty(max(cap_ev, 1), np.uint64), np.empty(max(cap_ev, 1), 'S1')
            w = 0
            for f5f, z, meta, n, eo, ne, ms, args, mv, ro in opened:
                mev_off = np.empty(n + 1, np.int64)
                # (the basecaller's mean / stdv are merged later and only if a read of the batch has an empty event: _fallback_values)
                margs = (n, min(len(a) for a in args + [mv, ms]), eo.ctypes.data, args[0].ctypes.data, args[1].ctypes.data, args[2].ctypes.data, args[3].ctypes.data,
                         ms.ctypes.data, ms.dtype.itemsize // 4, mv.ctypes.data, mev_off.ctypes.data)
                got = lib.dm_events_merge(*margs, None, None, m_start.ctypes.data + 8 * w, m_len.ctypes.data + 8 * w, m_base.ctypes.data + w)
                if got < 0:             # offsets that decrease or run past the table: a damaged container, the batch goes on without it
                    out.errors["Cannot open fast5 or other errors"].append(f5f)
                    print("Cannot open fast5 or other errors: {}".format(f5f))
                    continue
                merges.append((margs, ne, got, w, (eo, ms, mv, mev_off, args)))
                per_read = mev_off[1:] - mev_off[:-1]
                for i, m in enumerate(meta):
                    rid = m['read_id'].replace(" ", ":::").replace("\t", "|||")
                    if per_read[i] == 0:
                        out.errors['No events data'].append(f5f)
                        rid = None
                    ids.append(rid)
                    id_src.append(f5f)
                raw_parts.append(z['raw'][:int(ro[-1])])          # samples behind the last read's end would shift every later container's offsets
                raw_offs.extend((raw_offs[-1] + ro[1:]).tolist())
                ev_offs.extend((ev_offs[-1] + mev_off[1:]).tolist())
                w += got
            t1 = time.perf_counter()
            out.timing['load'] += t1 - t0
            if ids:
                cat = lambda parts, dt: np.concatenate(parts) if len(parts) > 1 else _c_arr(parts[0], dt)
                m_start, m_len, m_base = m_start[:w], m_len[:w], m_base[:w]
                m_mean = m_stdv = None

                def _fallback_values():
                    """The basecaller's mean / stdv of every merged event of the batch (getEvent's rounding): needed only for events at or behind a read's
                    first empty event - the containers are merged once more, this time with the value columns."""
                    mm_, ms_ = np.empty(max(w, 1), np.float32), np.empty(max(w, 1), np.float32)
                    for margs, ne, got, at, _keep in merges:
                        a_, b_ = np.empty(ne, np.float32), np.empty(ne, np.float32)
                        scratch = (np.empty(ne, np.uint64), np.empty(ne, np.uint64), np.empty(ne, 'S1'))
                        if lib.dm_events_merge(*margs, a_.ctypes.data, b_.ctypes.data, scratch[0].ctypes.data, scratch[1].ctypes.data, scratch[2].ctypes.data) != got:
                            raise _lib.DeepModHipError('dm_events_merge: ' + _lib.last_error())
                        mm_[at:at + got], ms_[at:at + got] = a_[:got], b_[:got]
                    return mm_[:w], ms_[:w]
                raw_off, mev_off = np.array(raw_offs, np.int64), np.array(ev_offs, np.int64)
                # resident form (round 6): a batch of raw containers only, whose rows are built on the device anyway - the signal request is POSTED
                # (samples + event tables into the server's request file, no wait) and its statistics never come back: the feeder needs only
                # first_empty (host arithmetic, dm_signal_plan_batch) to walk its alignments while the signal kernels run
                want_resident = (hasattr(normalizer, 'post_arrays') and len(raw_files) == len(files)
                                 and bool(moptions.get('select_base', os.environ.get('DEEPMOD_SELECT_BASE', '1') != '0'))
                                 and bool(moptions.get('rows_on_device', os.environ.get('DEEPMOD_ROWS_ON_DEVICE', '1') != '0'))
                                 and bool(moptions.get('stats_on_device', os.environ.get('DEEPMOD_STATS_ON_DEVICE', '1') != '0')))
                try:
                    if want_resident:
                        first_empty = np.empty(len(raw_off) - 1, np.int64)
                        _lib.check(lib.dm_signal_plan_batch(len(raw_off) - 1, raw_off.ctypes.data, mev_off.ctypes.data, m_start.ctypes.data, m_len.ctypes.data,
                                                            first_empty.ctypes.data))
                        if bool((first_empty < (mev_off[1:] - mev_off[:-1])).any()):
                            m_mean, m_stdv = _fallback_values()
                        out.sig = normalizer.post_arrays(raw_parts, raw_off, m_start, m_len, mev_off, first_empty, m_mean, m_stdv)
                        s_mean = s_stdv = None
                    else:
                        s_mean, s_stdv, first_empty = normalizer.event_stats_arrays(raw_parts, raw_off, m_start, m_len, mev_off)
                        if bool((np.asarray(first_empty) < (mev_off[1:] - mev_off[:-1])).any()):
                            m_mean, m_stdv = _fallback_values()
                except _lib.DeepModHipError:
                    # a read the batched signal call cannot take (events covering no signal): the per-read Python path reports it
                    lib.dm_rows_destroy(h)
                    h = None
                    return _prepare_batch_py(moptions, files, make_normalizer, alloc)
                t2 = time.perf_counter()
                out.timing['signal'] += t2 - t1
                # alignment records: the reference's own aligner call when the binary is on PATH, else the side-car .sam files
                f5data = {}
                for gi, rid in enumerate(ids):
                    if rid is None:
                        continue
                    if rid in f5data:
                        print('Duplicate id', rid, id_src[gi])
                    call = m_base[mev_off[gi]:mev_off[gi + 1]].tobytes().decode('ascii', 'replace') if moptions.get('Ref') else ''
                    f5data[rid] = (call, gi, None, id_src[gi], (0, 0))
                align_info = detect._alignment_lines(moptions, {'Error': out.errors}, raw_files, f5data)
                if align_info is None:
                    for f5k in sorted(f5data.keys()):
                        out.errors["Cannot running aligment"].append(f5data[f5k][3])
                else:
                    sp_param = {'f5data': f5data, 'ref_info': {}, 'f5status': "", 'line': ""}
                    f5align = readmap.parse_sam(moptions, {'Error': out.errors}, sp_param, align_info, f5data)
                    recs = list(f5align.items())
                    nrec = len(recs)
                    seqs = readmap.read_fasta(moptions['Ref']) if moptions.get('Ref') else {}
                    ref_bytes = _REF_BYTES.setdefault(moptions.get('Ref'), {})
                    flag = np.zeros(nrec, np.int32); pos1 = np.zeros(nrec, np.int64); rlen = np.zeros(nrec, np.int64)
                    cidx = np.full(nrec, -1, np.int32); ev_read = np.zeros(nrec, np.int32); skip = np.zeros(nrec, np.uint8)
                    cig_b, seq_b = [], []
                    for i, (qname, (mapq, fl, rname, ps, cigar, seq)) in enumerate(recs):
                        flag[i], pos1[i], ev_read[i] = fl, ps, f5data[qname][1]
                        cig_b.append(cigar.encode('ascii')); seq_b.append(seq.encode('ascii'))
                        rlen[i] = len(seq_b[-1])
                        if (not moptions.get('ConUnk', True)) and any(ch in rname for ch in '_-/:'):
                            skip[i] = 1
                        if rname not in contigs:
                            contigs[rname] = len(contigs)
                        cidx[i] = contigs[rname]
                        if rname in seqs and rname not in ref_bytes:
                            ref_bytes[rname] = seqs[rname].encode('ascii')
                        if rname not in seqs:
                            print('Fatal Error!!! cannot find the chrosome sequence %s' % rname)
                    names = sorted(contigs, key=contigs.get)
                    nct = len(names)
                    ref_ptr = (ctypes.c_char_p * max(nct, 1))(*[ref_bytes.get(nm) for nm in names])
                    ref_len = np.array([len(ref_bytes[nm]) if nm in ref_bytes else 0 for nm in names] or [0], np.int64)
                    cig_ptr = (ctypes.c_char_p * max(nrec, 1))(*cig_b)
                    seq_ptr = (ctypes.c_char_p * max(nrec, 1))(*seq_b)
                    region = [mr for mr in moptions.get('region', [[None, None, None]])]
                    any_all = any(mr[0] in ['', None] and mr[1] in ['', None] and mr[2] in ['', None] for mr in region)
                    if any_all:
                        region = []
                    elif not region:
                        skip[:] = 1          # an EMPTY region list matches nothing (myDetect.py:548-556 leaves isinreg False): n_region = 0 below means "no filter"
                    rg_c = np.array([(-1 if mr[0] in ['', None] else contigs.get(mr[0], 0x7fffffff)) for mr in region] or [0], np.int32)   # a contig no record of the batch names: matches nothing
                    rg_lo = np.array([(-1 if mr[1] in ['', None] else int(mr[1])) for mr in region] or [0], np.int64)
                    rg_hi = np.array([(-1 if mr[2] in ['', None] else int(mr[2])) for mr in region] or [0], np.int64)
                    keep.extend([flag, pos1, rlen, cidx, ev_read, skip, cig_b, seq_b, ref_ptr, ref_len, cig_ptr, seq_ptr, mev_off, m_mean, m_stdv,
                                 m_len, m_base, s_mean, s_stdv, first_empty, rg_c, rg_lo, rg_hi])
                    _lib.check(lib.dm_rows_add_raw(h, nrec, flag.ctypes.data, pos1.ctypes.data, cig_ptr, seq_ptr, rlen.ctypes.data, cidx.ctypes.data,
                                                   ev_read.ctypes.data, skip.ctypes.data, nct, ref_ptr, ref_len.ctypes.data, len(mev_off) - 1, len(m_len), mev_off.ctypes.data,
                                                   None if m_mean is None else m_mean.ctypes.data, None if m_stdv is None else m_stdv.ctypes.data,
                                                   m_len.ctypes.data, m_base.ctypes.data,
                                                   None if s_mean is None else s_mean.ctypes.data, None if s_stdv is None else s_stdv.ctypes.data,
                                                   first_empty.ctypes.data, len(region), rg_c.ctypes.data, rg_lo.ctypes.data, rg_hi.ctypes.data))
                    srcs.extend(f5data[q][3] for q, _ in recs)
                    for nm in names:
                        if nm in ref_bytes:
                            out.contig_len[nm] = len(ref_bytes[nm])
                out.timing['map+features'] += time.perf_counter() - t2
            t0 = time.perf_counter()
        for cf in files:
            if cf.endswith(rawreads.RAW_SUFFIX):
                continue
            try:
                pk = predstore.load_packed(cf)
            except Exception:
                out.errors["Cannot open container"].append(cf)
                continue
            t1 = time.perf_counter()
            out.timing['load'] += t1 - t0
            meta = pk['reads']
            n = len(meta)
            if n:
                for m in meta:
                    if m['chr'] not in contigs:
                        contigs[m['chr']] = len(contigs)
                arrs = [_c_arr(pk['row_off'], np.int64), _c_arr(pk['bmi_off'], np.int64), _c_arr(pk['ev_off'], np.int64), _c_arr(pk['tx'], np.float32),
                        _c_arr(pk['refbase'], 'S1'), _c_arr(pk['readbase'], 'S1'), _c_arr(pk['refbasei'], np.int64), _c_arr(pk['evbase'], 'S1'),
                        np.array([m['start_clip'] for m in meta], np.int64), np.array([m['end_clip'] for m in meta], np.int64),
                        np.array([contigs[m['chr']] for m in meta], np.int32), np.array([strands_c[m['strand']] for m in meta], np.int32)]
                keep.append(arrs)
                n_tab = min(len(arrs[4]), len(arrs[5]), len(arrs[6]))
                if (min(len(arrs[0]), len(arrs[1]), len(arrs[2])) != n + 1 or arrs[3].ndim != 2 or arrs[3].shape[1] != 7 or
                        lib.dm_rows_add_packed(h, n, len(arrs[3]), n_tab, len(arrs[7]), len(contigs), *[a.ctypes.data for a in arrs]) != 0):
                    # offset tables that do not fit their arrays (a truncated / damaged container): the file is reported, the batch goes on
                    out.errors["Cannot open container"].append(cf)
                    print("Cannot open container: %s (%s)" % (cf, _lib.last_error() or 'offset tables of the wrong length'))
                    continue
                srcs.extend([cf] * n)
            for c, ln in pk.get('contig_len', {}).items():
                out.contig_len[c] = max(out.contig_len.get(c, 0), int(ln))
            t0 = time.perf_counter()
            out.timing['rows'] += t0 - t1
        # ---- sizes, errors, then the arrays themselves (straight into the hand-over slot when alloc is given) ----
        nreads = len(srcs)
        info = np.zeros((max(nreads, 1), 8), np.int64)
        mism = np.zeros((20, 4), np.int64)
        R, T, S, nm = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        if lib.dm_rows_info(h, ctypes.byref(R), ctypes.byref(T), ctypes.byref(S), info.ctypes.data, mism.ctypes.data, 20, ctypes.byref(nm)) < 0:
            raise _lib.DeepModHipError("dm_rows_info: " + _lib.last_error())
        # compact form unless the caller wants every window classified (moptions['select_base'] = False / DEEPMOD_SELECT_BASE=0)
        compact = bool(moptions.get('select_base', os.environ.get('DEEPMOD_SELECT_BASE', '1') != '0'))
        for i in range(nreads):
            st = int(info[i, 0])
            if st in _ROWS_ERRORS:
                out.errors[_ROWS_ERRORS[st]].append(srcs[i])
            elif st == 3:
                print("Errorfast5 " + srcs[i])
                print('match-Error!!! no first and/or last match', srcs[i])
        for j in range(min(int(nm.value), 20)):
            print('Error Does not match: read %d of the batch (%s), table row %d, event %d, %d bases differ'
                  % (mism[j, 0], srcs[int(mism[j, 0])], mism[j, 1], mism[j, 2], mism[j, 3]))
        R, T, S = int(R.value), int(T.value), int(S.value)
        if compact:
            T = S + (T - R)                 # [S windows on a base of interest | extras]
        ok = info[:nreads, 0] == 0
        out.n_reads = int(ok.sum())
        out.n_windows = int(info[:nreads, 3][ok].sum())
        out.n_rows = R
        # device form (round 5): a batch of raw reads only hands over (mean, stdv, length) per event, a class byte per row and a descriptor per
        # read; the [R][7] matrix is built on the device.  moptions['rows_on_device'] = False / DEEPMOD_ROWS_ON_DEVICE=0: rows on the host as before
        dev_form = None
        resident = out.sig is not None
        if R and compact and (resident or bool(moptions.get('rows_on_device', os.environ.get('DEEPMOD_ROWS_ON_DEVICE', '1') != '0'))):
            ne, nr = ctypes.c_int64(), ctypes.c_int64()
            if lib.dm_rows_device_info(h, ctypes.byref(ne), ctypes.byref(nr)) == 1:
                dev_form = (0 if resident else int(ne.value), int(nr.value))
        if resident and R and dev_form is None:
            raise _lib.DeepModHipError('a batch whose statistics stay on the device must be a batch of raw reads')
        if R:
            if dev_form is not None:
                E, NR = dev_form
                if alloc is not None:
                    out.ev3, out.code, out.rdesc, out.pos, out.flags, sel = alloc(R, T, S, dev=dev_form)
                    out.sel = sel if S else np.zeros(0, np.int32)
                else:
                    out.ev3, out.code, out.rdesc = np.empty((max(E, 1), 3), np.float32), np.empty(R, np.uint8), np.empty((NR, 4), np.int64)
                    out.pos, out.flags, out.sel = np.empty(T, np.int64), np.empty(T, np.uint8), np.empty(S, np.int32)
                if resident:
                    out.ev3 = None
                out.rows = None
            elif alloc is not None:
                got = alloc(R, T, S) if compact else alloc(R, T)
                out.rows, out.pos, out.flags = got[:3]
                out.sel = (got[3] if S else np.zeros(0, np.int32)) if compact else None
            else:
                out.rows, out.pos, out.flags = np.empty((R, 7), np.float32), np.empty(T, np.int64), np.empty(T, np.uint8)
                out.sel = np.empty(S, np.int32) if compact else None
            names = sorted(contigs, key=contigs.get)
            rank = np.empty(max(len(names), 1), np.int32)
            rank[np.argsort(np.array(names, dtype=object), kind='stable') if names else []] = np.arange(len(names), dtype=np.int32)
            clen = np.zeros(max(len(names), 1), np.int64)
            groups = np.zeros((2 * max(len(names), 1), 8), np.int64)
            in_range = ctypes.c_int32(1)
            sel_dummy = np.zeros(1, np.int32)
            sel_ptr = (out.sel.ctypes.data if S else sel_dummy.ctypes.data) if compact else None
            if resident:
                ng = lib.dm_rows_emit_resident(h, rank.ctypes.data, out.code.ctypes.data, out.rdesc.ctypes.data, sel_ptr, out.pos.ctypes.data,
                                               out.flags.ctypes.data, groups.ctypes.data, len(groups), clen.ctypes.data, len(clen), ctypes.byref(in_range))
            elif dev_form is not None:
                ng = lib.dm_rows_emit_device(h, rank.ctypes.data, out.ev3.ctypes.data, out.code.ctypes.data, out.rdesc.ctypes.data, sel_ptr,
                                             out.pos.ctypes.data, out.flags.ctypes.data, groups.ctypes.data, len(groups), clen.ctypes.data, len(clen),
                                             ctypes.byref(in_range))
            else:
                ng = lib.dm_rows_emit(h, rank.ctypes.data, out.rows.ctypes.data, sel_ptr, out.pos.ctypes.data, out.flags.ctypes.data, groups.ctypes.data,
                                      len(groups), clen.ctypes.data, len(clen), ctypes.byref(in_range))
            if ng < 0:
                raise _lib.DeepModHipError("dm_rows_emit: " + _lib.last_error())
            out.groups = [(names[int(g[0])], '+-'[int(g[1])]) + tuple(int(v) for v in (g[2:8] if compact else g[2:6])) for g in groups[:ng]]
            for i, nmn in enumerate(names):
                if clen[i] > out.contig_len.get(nmn, 0):
                    out.contig_len[nmn] = int(clen[i])        # lower bound when no reference length is known
            out.f32 = not bool(in_range.value)
        out.timing['rows'] += time.perf_counter() - t0
        return out
    finally:
        if h:
            lib.dm_rows_destroy(h)
        del keep


def prepare_batch(moptions, files: List[str], make_normalizer=None, alloc=None) -> Prepared:
    """Host side of one worker batch (the reference's mDetect1 up to the call of mPredict1, myDetect.py:392-465, :488-715):
    raw containers go through signal normalisation, alignment records, dm_map_read and get_Feature; feature containers
    enter at the prediction step.  Default: the compiled path (_prepare_batch_c); moptions['rows_in_c'] = False (or
    DEEPMOD_ROWS_IN_C=0) selects the per-read Python restatement."""
    if moptions.get('rows_in_c', os.environ.get('DEEPMOD_ROWS_IN_C', '1') != '0'):
        return _prepare_batch_c(moptions, files, make_normalizer, alloc)
    return _prepare_batch_py(moptions, files, make_normalizer, alloc)


def _prepare_batch_py(moptions, files: List[str], make_normalizer=None, alloc=None) -> Prepared:
    """The per-read Python / numpy path (rounds 1-2): kept as the restatement the compiled path is tested against."""
    out = _PreparedBuilder()
    out.files = list(files)
    base = moptions['Base']
    t0 = time.perf_counter()
    raw_files = [f for f in files if f.endswith(rawreads.RAW_SUFFIX)]
    if raw_files:
        from . import detect, readmap
        sp_options = {'Error': out.errors}
        normalizer = make_normalizer() if make_normalizer else None
        f5data = rawreads.get_Event_Signals(moptions, sp_options, raw_files, normalizer)
        t1 = time.perf_counter()
        out.timing['signal'] += t1 - t0
        if f5data:
            align_info = detect._alignment_lines(moptions, sp_options, raw_files, f5data)
            if align_info is None:
                for f5k in sorted(f5data.keys()):
                    out.errors["Cannot running aligment"].append(f5data[f5k][3])
            else:
                sp_param = {'f5data': f5data, 'ref_info': {}, 'f5status': "", 'line': ""}
                f5align = readmap.parse_sam(moptions, sp_options, sp_param, align_info, f5data)
                reads = readmap.map_records(moptions, sp_options, sp_param, f5align, f5data)
                t2 = time.perf_counter()
                out.timing['map+features'] += t2 - t1
                rows_from_reads(reads, base, raw_files[0], out)
                for c, seq in sp_param['ref_info'].items():
                    out.contig_len[c] = len(seq)
                out.timing['rows'] += time.perf_counter() - t2
        t0 = time.perf_counter()
    for cf in files:
        if cf.endswith(rawreads.RAW_SUFFIX):
            continue
        try:
            pk = predstore.load_packed(cf)
        except Exception:
            out.errors["Cannot open container"].append(cf)
            continue
        t1 = time.perf_counter()
        out.timing['load'] += t1 - t0
        rows_from_packed(pk, base, cf, out)
        t0 = time.perf_counter()
        out.timing['rows'] += t0 - t1
    finish(out, alloc)
    out.timing['rows'] += time.perf_counter() - t0
    return out


# ---------------------------------------------------------------------------------------------
# feeder processes: the host side of a batch is numpy / zipfile / small C calls per read - Python threads serialise on the
# interpreter lock (32 feeder threads prepared 1.5x what one does), so the CLI prepares batches in worker PROCESSES and
# hands the three arrays over through a file in /dev/shm that the GPU process maps (and unlinks at once).
# ---------------------------------------------------------------------------------------------
_ALIGN = 256


def _shm_layout(n_rows: int, n_pos: int, n_sel: int = 0):
    """byte offsets of [rows f32[n_rows][7] | pos i64[n_pos] | flags u8[n_pos] | sel i32[n_sel]] -> (o_pos, o_flags, end, o_sel)"""
    o_pos = -(-n_rows * 28 // _ALIGN) * _ALIGN
    o_flags = o_pos + -(-n_pos * 8 // _ALIGN) * _ALIGN
    o_sel = o_flags + -(-max(n_pos, 1) // _ALIGN) * _ALIGN
    end = o_sel + 4 * n_sel if n_sel else o_flags + max(n_pos, 1)
    return o_pos, o_flags, end, o_sel


def _shm_views(buf, n_rows: int, n_pos: int, n_sel: int = 0):
    o_pos, o_flags, _, o_sel = _shm_layout(n_rows, n_pos, n_sel)
    out = (np.frombuffer(buf, np.float32, n_rows * 7, 0).reshape(n_rows, 7), np.frombuffer(buf, np.int64, n_pos, o_pos),
           np.frombuffer(buf, np.uint8, n_pos, o_flags))
    return out + (np.frombuffer(buf, np.int32, n_sel, o_sel),) if n_sel else out


def _shm_layout_dev(n_rows: int, n_pos: int, n_sel: int, n_ev: int, n_reads: int):
    """byte offsets of the device form [ev3 f32[n_ev][3] | code u8[n_rows] | rdesc i64[n_reads][4] | pos i64[n_pos] | flags u8[n_pos] | sel i32[n_sel]]
    -> dict(ev3, code, rdesc, pos, flags, sel, end)"""
    up = lambda v: -(-v // _ALIGN) * _ALIGN
    o = {'ev3': 0}
    o['code'] = up(12 * max(n_ev, 1))
    o['rdesc'] = o['code'] + up(max(n_rows, 1))
    o['pos'] = o['rdesc'] + up(32 * max(n_reads, 1))
    o['flags'] = o['pos'] + up(8 * max(n_pos, 1))
    o['sel'] = o['flags'] + up(max(n_pos, 1))
    o['end'] = o['sel'] + 4 * max(n_sel, 1)
    return o


def _shm_views_dev(buf, n_rows: int, n_pos: int, n_sel: int, n_ev: int, n_reads: int):
    o = _shm_layout_dev(n_rows, n_pos, n_sel, n_ev, n_reads)
    return (np.frombuffer(buf, np.float32, 3 * max(n_ev, 1), o['ev3']).reshape(-1, 3), np.frombuffer(buf, np.uint8, n_rows, o['code']),
            np.frombuffer(buf, np.int64, 4 * n_reads, o['rdesc']).reshape(n_reads, 4), np.frombuffer(buf, np.int64, n_pos, o['pos']),
            np.frombuffer(buf, np.uint8, n_pos, o['flags']), np.frombuffer(buf, np.int32, n_sel, o['sel']))


def usable_cpus() -> int:
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def shm_dir_for(moptions) -> str:
    """Directory of the hand-over files of this GPU process: /dev/shm when it has room (a container's default 64 MB tmpfs does
    not hold one batch), else the output folder (the files then go through the page cache of that file system)."""
    root = moptions['outFolder']
    try:
        st = os.statvfs('/dev/shm')
        if os.access('/dev/shm', os.W_OK) and st.f_bavail * st.f_frsize >= (2 << 30):
            root = '/dev/shm'
    except OSError:
        pass
    return os.path.join(root, 'deepmod_amd_%d' % os.getpid())


# ---------------------------------------------------------------------------------------------
# signal server: feeder processes do not own a HIP context.  The signal statistics of a raw-container batch
# (dm_signal_event_stats_batch) are computed by a thread of the GPU process: the feeder writes the samples and event tables
# of its batch into a shared-memory request file and waits for the answer.  (Every feeder with its own context capped a
# GPU at ~8 feeders: beyond that the hardware queues are oversubscribed and the classifier's queue is time-sliced.)
# ---------------------------------------------------------------------------------------------
def _sig_layout(n: int, n_raw: int, n_ev: int):
    """byte offsets of [raw i16 | raw_off i64 | ev_off i64 | ev_start u64 | ev_length u64 || mean f32 | stdv f32 | norm6 f64 | first_empty i64]"""
    up = lambda v: -(-v // 64) * 64
    o = {}
    pos = 0
    for name, nbytes in (('raw', 2 * n_raw), ('raw_off', 8 * (n + 1)), ('ev_off', 8 * (n + 1)), ('ev_start', 8 * n_ev), ('ev_length', 8 * n_ev)):
        o[name] = pos
        pos = up(pos + nbytes)
    o['in_end'] = pos
    for name, nbytes in (('mean', 4 * n_ev), ('stdv', 4 * n_ev), ('norm6', 48 * n), ('first_empty', 8 * n)):
        o[name] = pos
        pos = up(pos + nbytes)
    o['end'] = pos
    return o


def _sig_layout_res(n: int, n_raw: int, n_ev: int, with_fb: bool):
    """request of the resident form: [raw i16 | raw_off i64 | ev_off i64 | ev_start u64 | ev_length u64 | first_empty i64 | fb_mean f32 | fb_stdv f32]
    (the last two only when some read of the batch has an empty event) - inputs only: the statistics stay on the device"""
    up = lambda v: -(-v // 64) * 64
    o = {}
    pos = 0
    for name, nbytes in (('raw', 2 * n_raw), ('raw_off', 8 * (n + 1)), ('ev_off', 8 * (n + 1)), ('ev_start', 8 * n_ev), ('ev_length', 8 * n_ev),
                         ('first_empty', 8 * n), ('fb_mean', 4 * n_ev if with_fb else 0), ('fb_stdv', 4 * n_ev if with_fb else 0)):
        o[name] = pos
        pos = up(pos + nbytes)
    o['end'] = pos
    return o


class SignalResults:
    """GPU process: what the signal server threads hand to the batch loop - per resident request (feeder, number) the device block of its
    statistics, the range flag of dm_signal_event_stats_device and an error text.  The server registers a result when its call has returned (the block
    is then complete: nothing the batch loop queues afterwards needs a cross-stream wait); the loop takes it when the batch that names it arrives."""

    def __init__(self):
        self._cv = threading.Condition()
        self._done = {}

    def put(self, key, block, flags: int = 0, err: Optional[str] = None):
        with self._cv:
            self._done[tuple(key)] = (block, flags, err)
            self._cv.notify_all()

    def take(self, key, timeout: float = 300.0):
        key = tuple(key)
        end = time.perf_counter() + timeout
        with self._cv:
            while key not in self._done:
                left = end - time.perf_counter()
                if left <= 0:
                    raise RuntimeError('the signal stage never answered request %r' % (key,))
                self._cv.wait(left)
            return self._done.pop(key)

    def pending(self):
        with self._cv:
            return list(self._done)


class DeviceBlockPool:
    """Grow-only device blocks for the statistics of resident signal requests: taken by a server thread, given back by the batch loop once the launches
    that read a block have finished (its staging set's marker has passed)."""

    def __init__(self, device: int):
        self.device = device
        self._lock = threading.Lock()
        self._free = []
        self.allocated = 0

    def take(self, nbytes: int):
        from . import model as dm
        with self._lock:
            best = None
            for i, b in enumerate(self._free):
                if b.nbytes >= nbytes and (best is None or b.nbytes < self._free[best].nbytes):
                    best = i
            if best is not None:
                return self._free.pop(best)
        blk = dm.DeviceArray((int(nbytes * 1.25) + 4096,), np.uint8, self.device)
        with self._lock:
            self.allocated += 1
        return blk

    def give(self, blk):
        if blk is not None:
            with self._lock:
                self._free.append(blk)

    def close(self):
        with self._lock:
            blocks, self._free = self._free, []
        for b in blocks:
            b.free()


class RemoteSignalNormalizer:
    """Feeder-process side of the signal server: the interface of signal.SignalNormalizer that the raw path uses."""

    def __init__(self, wid: int, shm_dir: str, requests, answers):
        self.wid, self.requests, self.answers = wid, requests, answers
        self.path = os.path.join(shm_dir, 'sig_%d' % wid)
        self.mm = None
        self.size = 0
        self._posted = 0                   # resident requests posted so far
        self._acked = set()                # ... whose request file the server has copied (it may be written again)
        self._res_files = [None, None]     # (mapping, size, path) of the two request files of the resident form

    # ---- resident form: post and go on ----
    def post_arrays(self, raw_parts, raw_off, ev_start, ev_length, ev_off, first_empty, fb_mean=None, fb_stdv=None):
        """Write a request of the resident form (dm_signal_event_stats_device) into one of this feeder's two request files and queue it: no wait for the
        statistics - they stay on the device, the GPU process finds them under the returned (feeder, number) when the batch arrives.  The only wait is
        for the file itself: request k reuses the file of request k - 2, which the server must have copied into its page-locked memory (its
        acknowledgement; usually long there)."""
        import mmap
        from . import _lib
        seq = self._posted = self._posted + 1
        while seq - 2 > 0 and (seq - 2) not in self._acked:
            self._take_answer(block=True)
        self._acked.discard(seq - 2)
        n = len(raw_off) - 1
        n_raw, n_ev = int(raw_off[-1]), int(ev_off[-1])
        with_fb = fb_mean is not None and fb_stdv is not None
        o = _sig_layout_res(n, n_raw, n_ev, with_fb)
        files = self._res_files
        slot = seq % 2
        cur = files[slot]
        if cur is None or cur[1] < o['end']:
            if cur is not None:
                cur[0].close()
            size = max(1 << 22, int(o['end'] * 1.5))
            path = self.path + '_r%d' % slot
            fd = os.open(path, os.O_CREAT | os.O_RDWR, 0o600)
            try:
                os.ftruncate(fd, size)
                cur = files[slot] = (mmap.mmap(fd, size), size, path)
            finally:
                os.close(fd)
        mm, size, path = cur
        np.concatenate([np.asarray(p) for p in raw_parts], out=np.frombuffer(mm, np.int16, n_raw, o['raw']), casting='same_kind')
        np.frombuffer(mm, np.int64, n + 1, o['raw_off'])[:] = raw_off
        np.frombuffer(mm, np.int64, n + 1, o['ev_off'])[:] = ev_off
        np.frombuffer(mm, np.uint64, n_ev, o['ev_start'])[:] = ev_start
        np.frombuffer(mm, np.uint64, n_ev, o['ev_length'])[:] = ev_length
        np.frombuffer(mm, np.int64, n, o['first_empty'])[:] = first_empty
        if with_fb:
            np.frombuffer(mm, np.float32, n_ev, o['fb_mean'])[:] = fb_mean
            np.frombuffer(mm, np.float32, n_ev, o['fb_stdv'])[:] = fb_stdv
        self.requests.put(('res', self.wid, path, size, n, n_raw, n_ev, seq, with_fb))
        return (self.wid, seq)

    def _take_answer(self, block: bool = True):
        """One message from the server: ('ack', number, error) of a resident request, or the answer (None / error text) of a synchronous one."""
        from . import _lib
        msg = self.answers.get() if block else self.answers.get_nowait()
        if isinstance(msg, tuple) and msg and msg[0] == 'ack':
            if msg[2] is not None:
                raise _lib.DeepModHipError(msg[2])
            self._acked.add(msg[1])
            return 'ack'
        return ('answer', msg)

    def _wait_answer(self):
        while True:
            got = self._take_answer(block=True)
            if got != 'ack':
                return got[1]

    def _ensure(self, nbytes: int):
        import mmap
        if nbytes > self.size:
            if self.mm is not None:
                self.mm.close()
            size = max(1 << 22, int(nbytes * 1.5))
            fd = os.open(self.path, os.O_CREAT | os.O_RDWR, 0o600)
            try:
                os.ftruncate(fd, size)
                self.mm = mmap.mmap(fd, size)
            finally:
                os.close(fd)
            self.size = size

    def event_stats_batch(self, reads):
        from . import _lib
        if not reads:
            return []
        n = len(reads)
        raw_off = np.concatenate([[0], np.cumsum([len(r[0]) for r in reads])]).astype(np.int64)
        ev_off = np.concatenate([[0], np.cumsum([len(r[1]) for r in reads])]).astype(np.int64)
        n_raw, n_ev = int(raw_off[-1]), int(ev_off[-1])
        o = _sig_layout(n, n_raw, n_ev)
        self._ensure(o['end'])
        mm = self.mm
        np.concatenate([np.asarray(r[0]) for r in reads], out=np.frombuffer(mm, np.int16, n_raw, o['raw']), casting='same_kind')
        np.frombuffer(mm, np.int64, n + 1, o['raw_off'])[:] = raw_off
        np.frombuffer(mm, np.int64, n + 1, o['ev_off'])[:] = ev_off
        np.concatenate([np.asarray(r[1]) for r in reads], out=np.frombuffer(mm, np.uint64, n_ev, o['ev_start']), casting='same_kind')
        np.concatenate([np.asarray(r[2]) for r in reads], out=np.frombuffer(mm, np.uint64, n_ev, o['ev_length']), casting='same_kind')
        self.requests.put((self.wid, self.path, self.size, n, n_raw, n_ev))
        err = self._wait_answer()
        if err is not None:
            raise _lib.DeepModHipError(err)
        mean = np.frombuffer(mm, np.float32, n_ev, o['mean']).copy()
        stdv = np.frombuffer(mm, np.float32, n_ev, o['stdv']).copy()
        norm6 = np.frombuffer(mm, np.float64, 6 * n, o['norm6']).reshape(n, 6)
        first_empty = np.frombuffer(mm, np.int64, n, o['first_empty'])
        keys = ("mshift", "mscale", "read_med", "read_mad", "lower_lim", "upper_lim")
        return [(mean[ev_off[i]:ev_off[i + 1]], stdv[ev_off[i]:ev_off[i + 1]], dict(zip(keys, norm6[i].tolist())), int(first_empty[i]))
                for i in range(n)]

    def event_stats_arrays(self, raw_parts, raw_off, ev_start, ev_length, ev_off):
        """Same request as event_stats_batch, from arrays that are already back to back (see signal.SignalNormalizer.event_stats_arrays)."""
        from . import _lib
        n = len(raw_off) - 1
        n_raw, n_ev = int(raw_off[-1]), int(ev_off[-1])
        o = _sig_layout(n, n_raw, n_ev)
        self._ensure(o['end'])
        mm = self.mm
        np.concatenate([np.asarray(p) for p in raw_parts], out=np.frombuffer(mm, np.int16, n_raw, o['raw']), casting='same_kind')
        np.frombuffer(mm, np.int64, n + 1, o['raw_off'])[:] = raw_off
        np.frombuffer(mm, np.int64, n + 1, o['ev_off'])[:] = ev_off
        np.frombuffer(mm, np.uint64, n_ev, o['ev_start'])[:] = ev_start
        np.frombuffer(mm, np.uint64, n_ev, o['ev_length'])[:] = ev_length
        self.requests.put((self.wid, self.path, self.size, n, n_raw, n_ev))
        err = self._wait_answer()
        if err is not None:
            raise _lib.DeepModHipError(err)
        return (np.frombuffer(mm, np.float32, n_ev, o['mean']).copy(), np.frombuffer(mm, np.float32, n_ev, o['stdv']).copy(),
                np.frombuffer(mm, np.int64, n, o['first_empty']).copy())

    def event_stats(self, raw, ev_start, ev_length, want_signal: bool = False):
        if want_signal:
            raise ValueError('the signal server returns event statistics only')
        mean, stdv, norm, first_empty = self.event_stats_batch([(raw, ev_start, ev_length)])[0]
        return mean, stdv, norm, first_empty, None

    def close(self):
        if self.mm is not None:
            self.mm.close()
            self.mm = None
        for cur in self._res_files:
            if cur is not None:
                cur[0].close()
        self._res_files = [None, None]


def signal_server(requests, answers, device: int, stats=None, results: Optional[SignalResults] = None, blocks: Optional[DeviceBlockPool] = None):
    """Thread body in the GPU process: requests until None.
      (wid, path, file size, n, n_raw, n_ev)                          synchronous form: answers[wid] gets None or an error text; the statistics
                                                                       go back into the request file;
      ('res', wid, path, file size, n, n_raw, n_ev, number, with_fb)  resident form (round 6): answers[wid] gets ('ack', number, error) as soon
                                                                       as the request file has been copied (the feeder may write it again), the
                                                                       statistics go into a device block registered in `results` under (wid, number).
    Inputs are copied to page-locked memory (uploads from the shared-memory mapping itself are slow)."""
    import mmap
    from . import _lib, model as dm, signal as dmsignal
    norm = None           # its dm_signal handle is made for the first request: a run of feature containers never needs one, and
    lib = _lib.load()     # a raw run's first request arrives after the model is on the device (no three-way race for the HIP start-up)
    maps = {}
    pinned = None
    try:
        while True:
            req = requests.get()
            if req is None:
                return
            resident = req[0] == 'res'
            if resident:
                _, wid, path, size, n, n_raw, n_ev, seq, with_fb = req
            else:
                wid, path, size, n, n_raw, n_ev = req
            t0 = time.perf_counter()
            acked = False
            blk = None
            try:
                if norm is None:
                    norm = dmsignal.SignalNormalizer(device)
                if maps.get(path, (None, 0))[1] != size:
                    if path in maps:
                        maps[path][0].close()
                    fd = os.open(path, os.O_RDWR)
                    try:
                        maps[path] = (mmap.mmap(fd, size), size)
                    finally:
                        os.close(fd)
                mm = maps[path][0]
                o = _sig_layout_res(n, n_raw, n_ev, with_fb) if resident else _sig_layout(n, n_raw, n_ev)
                n_in = o['end'] if resident else o['in_end']
                if pinned is None or pinned.nbytes < o['end']:
                    if pinned is not None:
                        pinned.free()
                    pinned = dm.PinnedArray(int(o['end'] * 1.5) + 4096, device)
                pv = pinned.view(np.uint8, o['end'])
                pv[:n_in] = np.frombuffer(mm, np.uint8, n_in)
                base = pinned.ptr
                t1 = time.perf_counter()
                if resident:
                    answers[wid].put(('ack', seq, None))          # the request file is free again
                    acked = True
                    blk = blocks.take(12 * max(n_ev, 1))
                    flags = ctypes.c_int32(0)
                    rc = lib.dm_signal_event_stats_device(norm._h, n, base + o['raw'], base + o['raw_off'], base + o['ev_start'], base + o['ev_length'],
                                                          base + o['ev_off'], base + o['first_empty'], (base + o['fb_mean']) if with_fb else None,
                                                          (base + o['fb_stdv']) if with_fb else None, blk.ptr, None, ctypes.byref(flags))
                    if stats is not None:
                        stats['signal_server_call'] += time.perf_counter() - t1
                        stats['signal_server_copy'] += t1 - t0
                    if rc != 0:
                        results.put((wid, seq), None, 0, 'signal stage: ' + _lib.last_error())
                    else:
                        results.put((wid, seq), blk, int(flags.value), None)
                        blk = None              # the batch loop owns it now
                else:
                    rc = lib.dm_signal_event_stats_batch(norm._h, n, base + o['raw'], base + o['raw_off'], base + o['ev_start'], base + o['ev_length'],
                                                         base + o['ev_off'], base + o['mean'], base + o['stdv'], base + o['norm6'], base + o['first_empty'])
                    if stats is not None:
                        stats['signal_server_call'] += time.perf_counter() - t1
                    if rc != 0:
                        answers[wid].put(_lib.last_error())
                        continue
                    np.frombuffer(mm, np.uint8, o['end'] - o['in_end'], o['in_end'])[:] = pv[o['in_end']:]
                    answers[wid].put(None)
            except Exception as exc:                        # the feeder turns it into a per-read failure / the batch loop fails the run
                if resident:
                    if not acked:
                        answers[wid].put(('ack', seq, None))
                    results.put((wid, seq), None, 0, 'signal server: %r' % (exc,))
                else:
                    answers[wid].put('signal server: %r' % (exc,))
            finally:
                if blk is not None:             # a request that failed after taking its block
                    blocks.give(blk)
            if stats is not None:
                stats['signal_server'] += time.perf_counter() - t0
                stats['signal_requests'] += 1
                stats['signal_samples'] += n_raw
                stats['signal_events'] += n_ev
    finally:
        for mm, _ in maps.values():
            mm.close()
        if pinned is not None:
            pinned.free()
        if norm is not None:
            norm.close()



def feeder_process_main(moptions, work, ready, device: int, shm_dir: str, wid: int, free_slots=None, slot_bytes: int = 0,
                        sig_requests=None, sig_answers=None):
    """Body of one feeder process: file lists from `work` (a multiprocessing queue of (files, ...) items, filled before the
    run starts; empty = done) -> prepare_batch -> arrays into shared memory -> a small description on `ready`.
    Shared memory = one of the GPU process' page-locked slot files (`free_slots`: queue of slot numbers; the batch must fit
    `slot_bytes`), else a file of its own that the GPU process maps and unlinks."""
    import mmap
    import traceback
    norm = []

    def normalizer():                      # created on first use: only batches with raw containers need the signal stage
        if not norm:
            if sig_requests is not None:   # the GPU process' signal server: this process owns no HIP context
                norm.append(RemoteSignalNormalizer(wid, shm_dir, sig_requests, sig_answers))
            else:
                from . import signal as dmsignal
                norm.append(dmsignal.SignalNormalizer(device))
        return norm[0]

    slot_maps = {}

    def slot_map(i):
        if i not in slot_maps:
            fd = os.open(os.path.join(shm_dir, 'slot_%d' % i), os.O_RDWR)
            try:
                slot_maps[i] = mmap.mmap(fd, slot_bytes)
            finally:
                os.close(fd)
        return slot_maps[i]

    seq = 0
    try:
        while True:
            try:
                item = work.get(block=False)
            except Exception:              # queue.Empty (also through a manager proxy): nothing left
                break
            files = item[0] if isinstance(item, tuple) else item
            path = os.path.join(shm_dir, 'b%d_%d' % (wid, seq))
            seq += 1
            holder = {}

            def alloc(n_rows, n_pos, n_sel=0, dev=None):
                holder['dev'] = dev
                size = _shm_layout(n_rows, n_pos, n_sel)[2] if dev is None else _shm_layout_dev(n_rows, n_pos, n_sel, *dev)['end']
                views = (lambda b: _shm_views(b, n_rows, n_pos, n_sel)) if dev is None else (lambda b: _shm_views_dev(b, n_rows, n_pos, n_sel, *dev))
                if free_slots is not None and size <= slot_bytes:
                    holder['slot'] = free_slots.get()                    # blocks while the device queue holds every slot
                    return views(slot_map(holder['slot']))
                fd = os.open(path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
                try:
                    os.ftruncate(fd, size)
                    holder['mm'] = mmap.mmap(fd, size)
                finally:
                    os.close(fd)
                return views(holder['mm'])

            pb = prepare_batch(moptions, files, normalizer, alloc)
            meta = {'path': path if 'mm' in holder else None, 'slot': holder.get('slot'), 'n_rows': pb.n_rows, 'n_pos': len(pb.pos),
                    'groups': pb.groups, 'n_windows': pb.n_windows, 'n_reads': pb.n_reads, 'errors': {k: list(v) for k, v in pb.errors.items()},
                    'contig_len': dict(pb.contig_len), 'timing': dict(pb.timing), 'files': list(pb.files), 'f32': pb.f32,
                    'n_sel': None if pb.sel is None else len(pb.sel), 'dev': holder.get('dev') if (pb.ev3 is not None or pb.code is not None) else None,
                    'sig': pb.sig}
            pb.rows = pb.pos = pb.flags = pb.sel = pb.ev3 = pb.code = pb.rdesc = None          # drop the views before a mapping goes away
            if 'mm' in holder:
                holder['mm'].close()
            ready.put(meta)
    except BaseException:
        ready.put({'failed': traceback.format_exc()})
    finally:
        ready.put(None)


def prepared_from_shm(meta, slot_buffer=None) -> Prepared:
    """The GPU process' view of a batch a feeder process prepared: in one of its page-locked slots (slot_buffer(i) -> the
    slot's mapping), or in a file of its own (unlinked as soon as it is mapped)."""
    import mmap
    pb = Prepared()
    pb.n_rows, pb.groups, pb.n_windows, pb.n_reads = meta['n_rows'], [tuple(g) for g in meta['groups']], meta['n_windows'], meta['n_reads']
    for k, v in meta['errors'].items():
        pb.errors[k].extend(v)
    pb.contig_len = dict(meta['contig_len'])
    for k, v in meta['timing'].items():
        pb.timing[k] += v
    pb.files = meta['files']
    pb.f32 = bool(meta.get('f32', False))
    n_sel = meta.get('n_sel')
    dev = meta.get('dev')
    mk = (lambda b: _shm_views(b, meta['n_rows'], meta['n_pos'], n_sel or 0)) if dev is None else \
         (lambda b: _shm_views_dev(b, meta['n_rows'], meta['n_pos'], n_sel or 0, *dev))
    views = None
    if meta.get('slot') is not None:
        views = mk(slot_buffer(meta['slot']))
    elif meta['path'] is not None:
        fd = os.open(meta['path'], os.O_RDONLY)
        try:
            mm = mmap.mmap(fd, 0, prot=mmap.PROT_READ)
        finally:
            os.close(fd)
            os.unlink(meta['path'])
        views = mk(mm)
    pb.sig = tuple(meta['sig']) if meta.get('sig') is not None else None
    if views is not None and dev is not None:
        pb.ev3, pb.code, pb.rdesc, pb.pos, pb.flags = views[:5]
        if pb.sig is not None:             # resident form: the statistics are a device block of the signal stage, nothing per event in the slot
            pb.ev3 = None
        pb.rows = None
        pb.sel = views[5] if n_sel else np.zeros(0, np.int32)
        return pb
    if views is not None:
        pb.rows, pb.pos, pb.flags = views[:3]
    if n_sel is not None:
        pb.sel = views[3] if (views is not None and n_sel) else np.zeros(0, np.int32)
    return pb


# ---------------------------------------------------------------------------------------------
# device backend (C ABI) and the rank-local engine
# ---------------------------------------------------------------------------------------------
class HipBackend:
    """The device side of the engine through libdeepmod_hip.so: one model, one in-order stream, NSET staging sets (page-locked
    host memory + device buffers, grow-only), one PositionSummary per contig x strand.  Batch k is copied into set k % NSET
    while the device works on the batches before it; a stream marker per set says when the set may be refilled."""
    NSET = 3

    def __init__(self, moptions, device: int):
        from . import _lib, model as dm
        self._lib = _lib.load()
        self.device = device
        _, init_l, _, _, _, X, Y, _, _, _, _, mfpred = dm.mCreateSession(moptions['fnum'], moptions['hidden'], moptions['windowsize'], moptions)
        self.sess = dm.new_session(device)
        dm.import_meta_graph(moptions['modfile'][0] + '.meta').restore(
            self.sess, dm.latest_checkpoint(moptions['modfile'][1]) or moptions['modfile'][0])
        self.model = self.sess.model
        self.model.set_option(_lib.DM_OPT_ASYNC, 1)
        self._lib_mod = _lib
        self._precision = self.model.get_info(_lib.DM_INFO_PRECISION)
        # CUs left to the signal server's kernels (DM_OPT_RESERVED_CUS).  Measured: behind a classifier launch that holds every CU
        # the histogram kernel of a 40-read request takes 1.9 ms instead of 0.15 ms, but with 32 CUs reserved the classifier is
        # 4.5 % slower and the end-to-end rate no better (profiles/r02/README.md): default 0
        self.model.set_option(_lib.DM_OPT_RESERVED_CUS, int(moptions.get('reserved_cus', os.environ.get('DEEPMOD_RESERVED_CUS', 0))))
        self._dm = dm
        self._sets = [{'host': None, 'dev': None, 'sig_block': None} for _ in range(self.NSET)]
        # resident form: set by the engine when signal server threads run in this process (stream.SignalResults / DeviceBlockPool)
        self.signal_results = None
        self.signal_blocks = None
        self._k = 0
        self._h2d = self._lib.dm_model_h2d_ahead if int(os.environ.get('DEEPMOD_COPY_AHEAD', 1)) else self._lib.dm_model_h2d_async
        self.timing = defaultdict(float)   # where submit() spends the GPU process' time

    def _staging(self, st, nbytes):
        """(pinned host block, device block) of one set, both at least nbytes (the set's marker has passed: nothing reads them)."""
        if st['host'] is None or st['host'].nbytes < nbytes:
            cap = int(nbytes * 1.25) + 4096
            for blk in (st['host'], st['dev']):
                if blk is not None:
                    blk.free()
            st['host'] = self._dm.PinnedArray(cap, self.device)
            st['dev'] = self._dm.DeviceArray((cap,), np.uint8, self.device)
        return st['host'], st['dev']

    def new_summary(self, length: int):
        from . import summary
        s = summary.PositionSummary(length, self.device)
        s.follow(self.model)
        return s

    def submit(self, pb: Prepared, summaries) -> None:
        """Queue one batch: stage, upload, classify every row (windows assembled on the device), accumulate per group.
        Returns after enqueue; the batch's own arrays are free again on return (pb.on_done is called)."""
        resident = pb.sig is not None              # the statistics of the batch's events are a device block of the signal stage
        sig_block = None
        if resident:
            if self.signal_results is None:
                raise RuntimeError('a batch in the resident form reached a backend without a signal stage')
            t0 = time.perf_counter()
            sig_block, sig_flags, sig_err = self.signal_results.take(pb.sig)
            self.timing['wait_signal'] += time.perf_counter() - t0
            if sig_err is not None:
                raise self._lib_mod.DeepModHipError(sig_err)
            if sig_flags & 1:
                pb.f32 = True
            if pb.n_rows and len(pb.rdesc) and (int(pb.rdesc[:, 2].min()) < 0 or 12 * int(pb.rdesc[:, 3].max()) > sig_block.nbytes):
                # (the descriptors and the request come from the same feeder call; a mismatch would be a read outside the block on the device)
                self.signal_blocks.give(sig_block)
                raise RuntimeError('a batch whose read descriptors point outside the statistics of its signal request %r' % (pb.sig,))
        if pb.n_rows == 0:
            if sig_block is not None:
                self.signal_blocks.give(sig_block)
            if pb.on_done is not None:
                pb.on_done()
                pb.on_done = None
            return
        R, T = pb.n_rows, len(pb.pos)
        S = 0 if pb.sel is None else len(pb.sel)
        dev_form = pb.ev3 is not None or resident  # raw reads in the device form: the feature rows are built on the device (dm_rows_assemble)
        if dev_form:
            E, NR = (0 if resident else len(pb.ev3)), len(pb.rdesc)
            o = _shm_layout_dev(R, T, S, E, NR)
            o_pos, o_flags, end, o_sel = o['pos'], o['flags'], o['end'], o['sel']
            o_rows = -(-end // _ALIGN) * _ALIGN      # device only: nothing is uploaded behind `end`
            o_cls = o_rows + -(-R * 28 // _ALIGN) * _ALIGN
        else:
            o_pos, o_flags, end, o_sel = _shm_layout(R, T, S)
            o_rows = 0
            o_cls = -(-end // _ALIGN) * _ALIGN
        n_cls = R if pb.sel is None else max(S, 1)
        t0 = time.perf_counter()
        i = self._k % self.NSET
        self._k += 1
        self.model.wait_mark(i)                    # the launches that read this set (batch k - NSET) are done
        if self._sets[i]['sig_block'] is not None: # ... and with them the statistics block that batch read
            self.signal_blocks.give(self._sets[i]['sig_block'])
        self._sets[i]['sig_block'] = sig_block
        t1 = time.perf_counter()
        host, dev = self._staging(self._sets[i], o_cls + n_cls)
        if dev_form:
            if not resident:
                np.copyto(host.view(np.float32, 3 * E, o['ev3']).reshape(E, 3), pb.ev3, casting='same_kind')
            np.copyto(host.view(np.uint8, R, o['code']), pb.code, casting='same_kind')
            np.copyto(host.view(np.int64, 4 * NR, o['rdesc']).reshape(NR, 4), pb.rdesc, casting='same_kind')
        else:
            np.copyto(host.view(np.float32, R * 7).reshape(R, 7), pb.rows, casting='same_kind')
        np.copyto(host.view(np.int64, T, o_pos), pb.pos, casting='same_kind')
        np.copyto(host.view(np.uint8, T, o_flags), pb.flags, casting='same_kind')
        if S:
            np.copyto(host.view(np.int32, S, o_sel), pb.sel, casting='same_kind')
        if pb.on_done is not None:                 # e.g. the feeder's shared-memory slot goes back to its queue
            pb.on_done()
            pb.on_done = None
        t2 = time.perf_counter()
        # on the copy stream: the upload of this batch runs while the device classifies the batches before it (nothing queued reads
        # this set: its marker has passed), and the launches below wait for it
        self._lib_check(self._h2d(self.model._h, dev.ptr, host.ptr, end))
        d_rows, d_pos, d_flags, d_cls, d_sel = dev.ptr + o_rows, dev.ptr + o_pos, dev.ptr + o_flags, dev.ptr + o_cls, dev.ptr + o_sel
        if dev_form:
            self.model.assemble_rows_device(d_rows, dev.ptr + o['code'], sig_block.ptr if resident else dev.ptr + o['ev3'], dev.ptr + o['rdesc'], NR, R)
            self.timing['rows_on_device'] += R
            if resident:
                self.timing['stats_on_device'] += R
        if pb.sel is None:
            # classic form: window centred on row r -> cls[r]; rows 0..9 and R-10..R-1 are padding of the first / last read
            classify = lambda: self.model.predict_rows_device(d_rows, R, HALF, R - 2 * HALF, d_cls + HALF)
        else:
            # compact form: only the windows centred on a base of interest, cls[i] belongs to window sel[i]
            classify = lambda: (self.model.predict_rows_at_device(d_rows, R, d_sel, S, d_cls) if S else None)
        if pb.f32 and self._precision != self._lib_mod.DM_PREC_F32:      # launches are queued in order: the switch covers this batch only
            self.model.set_option(self._lib_mod.DM_OPT_PRECISION, self._lib_mod.DM_PREC_F32)
            classify()
            self.model.set_option(self._lib_mod.DM_OPT_PRECISION, self._precision)
            self.timing['f32_batches'] += 1
        else:
            classify()
        self.timing['classified'] += (R - 2 * HALF) if pb.sel is None else S
        xbase = R if pb.sel is None else S
        for g in pb.groups:
            c, s, lo, hi, xlo, xhi = g[:6]
            summ = summaries(c, s, pb.contig_len.get(c, 0))
            if pb.sel is None:
                summ.add_classified_device(d_pos + 8 * lo, d_flags + lo, d_cls + lo, hi - lo)
            elif g[7] > g[6]:
                summ.add_classified_device(d_pos + 8 * g[6], d_flags + g[6], d_cls + g[6], g[7] - g[6])
            if xhi > xlo:
                summ.add_device(d_pos + 8 * (xbase + xlo), d_flags + xbase + xlo, xhi - xlo)
        self.model.mark(i)
        t3 = time.perf_counter()
        self.timing['wait_device'] += t1 - t0
        self.timing['stage'] += t2 - t1
        self.timing['launch'] += t3 - t2

    def _lib_check(self, rc):
        from . import _lib
        _lib.check(rc)

    def sync(self):
        self.model.sync()

    def close(self):
        self.sync()
        for st in self._sets:
            for blk in (st['host'], st['dev']):
                if blk is not None:
                    blk.free()
            if st['sig_block'] is not None and self.signal_blocks is not None:
                self.signal_blocks.give(st['sig_block'])
            st['host'] = st['dev'] = st['sig_block'] = None
        if self.signal_blocks is not None:
            if self.signal_results is not None:         # statistics whose batch never arrived (a feeder that failed after posting its request)
                for key in self.signal_results.pending():
                    self.signal_blocks.give(self.signal_results.take(key, timeout=1.0)[0])
            self.signal_blocks.close()
        self.sess.close()


def finalize_threads(n_tables: int, longest: int, budget_positions: int = 250_000_000) -> int:
    """Tables fetched and formatted side by side by a single rank's finalize: bounded by the positions in flight."""
    return max(1, min(8, n_tables, budget_positions // max(int(longest), 1)))


class StreamEngine:
    """Rank-local streaming detect: pulls worker batches, keeps per contig x strand counters, merges at the end."""

    def __init__(self, moptions, backend, rank: int = 0, world: int = 1):
        self.mo, self.backend, self.rank, self.world = moptions, backend, rank, world
        self.summaries = {}
        self.ref_len: Dict[str, int] = {}
        self.errors: Dict[str, List[str]] = defaultdict(list)
        self.stats = defaultdict(float)

    def _summary(self, chrom: str, strand: str, need_len: int):
        """The counters of one contig x strand, at least need_len positions long.  With a reference length (FASTA or
        container metadata) they are allocated once; otherwise they grow geometrically as reads reach further."""
        key = (chrom, strand)
        want = max(self.ref_len.get(chrom, 0), need_len, 1)
        s = self.summaries.get(key)
        if s is None:
            s = self.summaries[key] = self.backend.new_summary(want)
        elif s.length < want:
            s.grow(want if chrom in self.ref_len else int(want * 1.5))
        return s

    def set_reference_lengths(self, lengths: Dict[str, int]):
        for c, ln in lengths.items():
            self.ref_len[c] = max(self.ref_len.get(c, 0), int(ln))

    def consume(self, pb: Prepared):
        t0 = time.perf_counter()
        for k, v in pb.errors.items():
            self.errors[k].extend(v)
        self.backend.submit(pb, self._summary)
        self.stats['submit'] += time.perf_counter() - t0
        self.stats['windows'] += pb.n_windows
        self.stats['rows'] += pb.n_rows
        self.stats['reads'] += pb.n_reads
        self.stats['batches'] += 1
        for k, v in pb.timing.items():
            self.stats['prep_' + k] += v

    def run(self, batches: Iterable, feeders: int = 2, make_normalizer=None):
        """batches: iterable of file lists (a shared queue drained by all ranks, or this rank's static shard).
        `feeders` threads prepare batches ahead of the device queue (numpy / zlib / the C ABI release the GIL)."""
        t_start = time.perf_counter()
        it = iter(batches)
        lock = threading.Lock()
        ready: "queue.Queue" = queue.Queue(maxsize=max(2, feeders))
        tl = threading.local()

        def normalizer():
            if make_normalizer is None:
                return None
            if not hasattr(tl, 'norm'):
                tl.norm = make_normalizer()
            return tl.norm

        def feed():
            while True:
                with lock:
                    try:
                        files = next(it)
                    except StopIteration:
                        files = None
                if files is None:
                    ready.put(None)
                    return
                try:
                    ready.put(prepare_batch(self.mo, files, normalizer))
                except BaseException as exc:      # surfaces in the consumer: a failed batch must not be dropped silently
                    ready.put(exc)
                    return

        threads = [threading.Thread(target=feed, daemon=True) for _ in range(max(1, feeders))]
        for t in threads:
            t.start()
        done = 0
        while done < len(threads):
            t0 = time.perf_counter()
            item = ready.get()
            self.stats['wait_feed'] += time.perf_counter() - t0
            if item is None:
                done += 1
            elif isinstance(item, BaseException):
                raise item
            else:
                self.consume(item)
        t0 = time.perf_counter()
        self.backend.sync()
        self.stats['drain'] += time.perf_counter() - t0
        self.stats['detect_wall'] += time.perf_counter() - t_start
        for k, v in getattr(self.backend, 'timing', {}).items():
            self.stats['submit_' + k] += v

    def run_processes(self, work, n_procs: int, device: int, ctx=None, make_backend=None):
        """Same as run(), with the batches prepared by `n_procs` feeder processes that drain `work` (a multiprocessing queue
        of (files, ...) items shared by all ranks and filled before the run starts).  make_backend: create the device backend
        only after the feeders have been started, so that they prepare the first batches while the model loads."""
        import multiprocessing
        import shutil
        t_start = time.perf_counter()
        ctx = ctx or multiprocessing.get_context('spawn')
        shm_dir = shm_dir_for(self.mo)
        os.makedirs(shm_dir, exist_ok=True)
        ready = ctx.Queue(maxsize=max(4, 2 * n_procs))
        # hand-over slots: files in shm_dir, mapped once by both sides, that the feeders fill and this process copies into its
        # page-locked staging memory (an upload straight from a shared-memory mapping ran at 0.7 GB/s, and page-locking the
        # mapping itself faulted the GPU).  A batch larger than a slot falls back to a file of its own.
        import mmap
        slot_bytes = int(self.mo.get('feeder_slot_mb', 128)) << 20
        n_slots = n_procs + 4 if self.mo.get('feeder_slots', True) else 0
        try:
            st = os.statvfs(shm_dir)
            n_slots = min(n_slots, int(0.5 * st.f_bavail * st.f_frsize) // slot_bytes)
        except OSError:
            n_slots = 0
        slots, free_slots = {}, None
        if n_slots >= 4:
            free_slots = ctx.Queue()
            for i in range(n_slots):
                fd = os.open(os.path.join(shm_dir, 'slot_%d' % i), os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
                os.ftruncate(fd, slot_bytes)
                os.close(fd)
                free_slots.put(i)

        def slot_buffer(i):                  # mapped on first use
            if i not in slots:
                fd = os.open(os.path.join(shm_dir, 'slot_%d' % i), os.O_RDWR)
                try:
                    slots[i] = mmap.mmap(fd, slot_bytes)
                finally:
                    os.close(fd)
            return slots[i]

        sig_requests = sig_answers = server = None
        if self.mo.get('signal_server', True) and (make_backend is not None or hasattr(self.backend, 'device')):
            sig_requests = ctx.Queue()
            sig_answers = [ctx.Queue() for _ in range(n_procs)]
            sig_results, sig_blocks = SignalResults(), DeviceBlockPool(device)
            server = [threading.Thread(target=signal_server, args=(sig_requests, sig_answers, device, self.stats, sig_results, sig_blocks), daemon=True)
                      for _ in range(max(1, int(self.mo.get('signal_servers', os.environ.get('DEEPMOD_SIGNAL_SERVERS', 2)))))]     # each with its own dm_signal handle / stream
            for th in server:
                th.start()
        procs = [ctx.Process(target=feeder_process_main, daemon=True,
                             args=(self.mo, work, ready, device, shm_dir, i, free_slots, slot_bytes, sig_requests,
                                   sig_answers[i] if sig_answers else None)) for i in range(n_procs)]
        for pr in procs:
            pr.start()
        self.stats['at_feeders_started'] = time.perf_counter() - t_start
        clean = False
        try:
            if make_backend is not None:
                t0 = time.perf_counter()
                self.backend = make_backend()
                self.stats['backend_init'] += time.perf_counter() - t0
            if server is not None and hasattr(self.backend, 'signal_results'):
                self.backend.signal_results, self.backend.signal_blocks = sig_results, sig_blocks
            self.stats['at_backend_ready'] = time.perf_counter() - t_start
            done = 0
            while done < n_procs:
                t0 = time.perf_counter()
                try:
                    item = ready.get(timeout=5.0)
                except queue.Empty:
                    if not any(pr.is_alive() for pr in procs) and ready.empty():
                        raise RuntimeError('feeder processes ended without finishing their batches (exit codes %s)'
                                           % [pr.exitcode for pr in procs])
                    self.stats['wait_feed'] += time.perf_counter() - t0
                    continue
                self.stats['wait_feed'] += time.perf_counter() - t0
                if item is None:
                    done += 1
                elif 'failed' in item:
                    raise RuntimeError('a feeder process failed:\n' + item['failed'])
                else:
                    self.stats.setdefault('at_first_batch', time.perf_counter() - t_start)
                    pb = prepared_from_shm(item, slot_buffer)
                    if item.get('slot') is not None:
                        pb.on_done = (lambda i=item['slot']: free_slots.put(i))
                        self.stats['slot_batches'] += 1
                    self.consume(pb)
            t0 = time.perf_counter()
            self.stats['at_last_batch'] = t0 - t_start
            self.backend.sync()
            self.stats['drain'] += time.perf_counter() - t0
            self.stats['at_drained'] = time.perf_counter() - t_start
            clean = True
        finally:
            for pr in procs:
                if clean:                # every feeder has sent its end marker: it is on its way out
                    pr.join(timeout=10)
                if pr.is_alive():        # the run is failing (or a feeder hangs): feeders may be blocked on a queue for good
                    pr.terminate()
            if server is not None:
                for _ in server:
                    sig_requests.put(None)
                for th in server:
                    th.join(timeout=10)
            shutil.rmtree(shm_dir, ignore_errors=True)
        self.stats['at_feeders_gone'] = time.perf_counter() - t_start
        self.stats['detect_wall'] += time.perf_counter() - t_start
        for k, v in getattr(self.backend, 'timing', {}).items():
            self.stats['submit_' + k] += v

    def finalize(self, gather, scatter_fn=None, write: bool = True, reduce_fn=None) -> Dict[Tuple[str, str], bytes]:
        """Merge over ranks and write the BED files.
        gather(obj) -> list of every rank's obj (control plane).  Data plane, world > 1:
          * scatter_fn(summary) -> (first, count): the summary's counters are summed over the ranks and THIS rank is left with the
            slice [first, first + count) (dm_summary_reduce_scatter).  Every rank fetches and formats its slice and writes it as a
            part file; rank 0 concatenates the parts in rank order (BED lines are sorted by position, so that IS the file).  No rank
            ever holds, downloads or formats a whole contig - for a human genome that is 74 GB of counters and 6e8 lines.
          * reduce_fn(summary) (fallback when no scatter_fn is given): sum into rank 0, which fetches and formats everything.
        All ranks walk the union of contig x strand keys in the same order.  Returns the BED bytes rank 0 assembled."""
        from . import summary as dmsum
        t0 = time.perf_counter()
        mine = {"%s\t%s" % k: self.summaries[k].length for k in self.summaries}
        lens = dict(self.ref_len)
        everyone = gather({"keys": mine, "len": lens}) if self.world > 1 else [{"keys": mine, "len": lens}]
        keys = sorted(set(k for e in everyone for k in e["keys"]))
        beds, parts = {}, {}
        # (moptions['force_scatter_merge']: the scatter form on a single rank too - how the GPU tests run this code path on one GPU)
        scattered = scatter_fn is not None and (self.world > 1 or bool(self.mo.get('force_scatter_merge')))
        out_path = lambda chrom, strand: '%s/mod_pos.%s%s.%s.bed' % (self.mo['outFolder'], chrom, strand, self.mo['Base'])
        def fetch_format_write(key):             # one rank, one table: download the counters, format, write (all outside the interpreter lock)
            chrom, strand = key.split("\t")
            s = self.summaries[(chrom, strand)]
            touch, cov, mod = s.fetch()
            s.close()
            if not write:
                return dmsum.bed_lines(chrom, strand, self.mo['Base'], touch, cov, mod), 0
            # slices of the table formatted side by side and written in order; the text of a large contig is never held whole
            fh, nbytes = None, 0
            for part in dmsum.bed_parts(chrom, strand, self.mo['Base'], touch, cov, mod):
                if fh is None:                      # the reference writes no file for an empty table (myDetect.py:1109)
                    fh = open(out_path(chrom, strand), 'wb')
                fh.write(part)
                nbytes += len(part)                 # (summed by the caller: several of these run side by side)
            if fh is not None:
                fh.close()
            return (b'' if fh is None else None), nbytes      # None: written to its file

        if not scattered and self.world == 1 and len(keys) >= 1:
            # a single rank: the tables are independent (own counters, own stream, own file) - contig x strand tables side by side
            from concurrent.futures import ThreadPoolExecutor
            for key in keys:
                chrom, strand = key.split("\t")
                self.summaries[(chrom, strand)].grow(max(self.summaries[(chrom, strand)].length, lens.get(chrom, 0)))
            # a table in flight holds 3 x length int32 on the host plus its text: at most ~2.5e8 positions (3 GB of counters) at a time,
            # i.e. eight E. coli-sized tables side by side but one chr1-sized table after the other
            longest = max(self.summaries[tuple(key.split("\t"))].length for key in keys)
            with ThreadPoolExecutor(finalize_threads(len(keys), longest)) as pool:
                for key, (bed, nbytes) in zip(keys, pool.map(fetch_format_write, keys)):
                    beds[tuple(key.split("\t"))] = bed
                    self.stats['bed_bytes'] += nbytes
            keys = []
        for key in keys:
            chrom, strand = key.split("\t")
            length = max([e["keys"].get(key, 0) for e in everyone] + [e["len"].get(chrom, 0) for e in everyone])
            s = self.summaries.get((chrom, strand))
            if s is None:                               # this rank saw no read of that contig x strand: zeros
                s = self.summaries[(chrom, strand)] = self.backend.new_summary(length)
            s.grow(length)                              # exactly the common length (no growth slack): equal counts on every rank
            if scattered:
                s.sync()                                # every rank: positions this rank dropped as out of range fail the run here
                first, count = scatter_fn(s)
                touch, cov, mod = s.fetch_slice()
                if write:
                    parts[key] = 0
                    with open(out_path(chrom, strand) + '.part%d' % self.rank, 'wb') as fh:
                        for part in dmsum.bed_parts(chrom, strand, self.mo['Base'], touch, cov, mod, first_pos=first):
                            fh.write(part)
                            parts[key] += len(part)
                else:
                    part = dmsum.bed_lines(chrom, strand, self.mo['Base'], touch, cov, mod, first_pos=first)
                    parts[key] = len(part)
                    beds[(chrom, strand)] = part        # (tests: the caller joins the ranks' parts itself)
                self.stats['bed_bytes'] += parts[key]
            else:
                if self.world > 1:
                    s.sync()
                    reduce_fn(s)
                if self.rank == 0:
                    touch, cov, mod = s.fetch()
                    bed = dmsum.bed_lines(chrom, strand, self.mo['Base'], touch, cov, mod)
                    beds[(chrom, strand)] = bed
                    if write and len(bed) > 0:          # the reference writes no file for an empty table (myDetect.py:1109)
                        with open(out_path(chrom, strand), 'wb') as fh:
                            fh.write(bed)
            s.close()
        self.summaries = {}
        if scattered and write:
            # every rank's parts are on disk once its sizes have been gathered; rank 0 joins them
            sizes = gather({"parts": parts}) if self.world > 1 else [{"parts": parts}]
            if self.rank == 0:
                for key in keys:
                    chrom, strand = key.split("\t")
                    names = [out_path(chrom, strand) + '.part%d' % r for r in range(self.world)]
                    total = sum(e["parts"].get(key, 0) for e in sizes)
                    if total > 0:                       # the reference writes no file for an empty table (myDetect.py:1109)
                        with open(out_path(chrom, strand), 'wb') as out:
                            for r, nm in enumerate(names):
                                with open(nm, 'rb') as fh:
                                    data = fh.read()
                                if len(data) != sizes[r]["parts"].get(key, 0):
                                    raise RuntimeError('BED part %s has %d bytes, its rank reported %d' % (nm, len(data), sizes[r]["parts"].get(key, 0)))
                                out.write(data)
                            beds[(chrom, strand)] = None
                    for nm in names:
                        if os.path.exists(nm):
                            os.remove(nm)
        self.stats['merge+bed'] += time.perf_counter() - t0
        return beds


# ---------------------------------------------------------------------------------------------
# process entry: one rank = one GPU
# ---------------------------------------------------------------------------------------------
class WorkList:
    """The work items of one run, known before its processes start, handed out through one shared counter: what the feeders
    and ranks of a streaming run take their batches from.  Same calls as the reference's h5files_Q (`get(block=False)`, queue.Empty
    when nothing is left), without a manager process in between (0.2 s of start-up and shut-down on a 2 s run) and without the
    window in which a multiprocessing.Queue that was filled a moment ago still reads as empty.
    The items travel to the processes through a file, not through their start-up pipe: a spawned process receives its arguments
    through a pipe of 64 KB, and a parent that writes more blocks until the child's interpreter is up and reading - four feeders
    with the 2,000 paths of a run in their arguments started one after the other (0.07 s each) instead of side by side."""

    def __init__(self, items, ctx, directory: Optional[str] = None):
        import pickle
        import tempfile
        self._items = list(items)
        self.n = len(self._items)
        self.head = ctx.Value('q', 0)
        if directory is None and os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK):
            directory = '/dev/shm'
        try:
            fd, self.path = tempfile.mkstemp(prefix='deepmod_work_', suffix='.pkl', dir=directory)
        except OSError:                      # (a full or read-only /dev/shm: the default temporary directory)
            fd, self.path = tempfile.mkstemp(prefix='deepmod_work_', suffix='.pkl')
        with os.fdopen(fd, 'wb') as fh:
            pickle.dump(self._items, fh, protocol=4)

    def __getstate__(self):
        return {'_items': None, 'n': self.n, 'head': self.head, 'path': self.path}

    def get(self, block: bool = False):
        with self.head.get_lock():
            i = self.head.value
            if i >= self.n:
                raise queue.Empty
            self.head.value = i + 1
        if self._items is None:
            import pickle
            with open(self.path, 'rb') as fh:
                self._items = pickle.load(fh)
        return self._items[i]

    def empty(self) -> bool:
        return self.head.value >= self.n

    def close(self):
        """By the process that made the list, once every process that takes from it has ended."""
        try:
            os.remove(self.path)
        except OSError:
            pass


def _drain(q):
    while True:
        try:
            item = q.get(block=False)
        except Exception:
            return
        yield item[0]


def stream_rank_main(moptions, rank: int, world: int, device: int, work, result_q=None, feeders: int = 2, feeder_procs: int = 0):
    """Body of one GPU process.  `work`: a shared queue of (files, subfolder, batchid) items drained by all ranks,
    or a list of file lists (static shard).  feeder_procs > 0 (queue only): batches are prepared by that many feeder
    processes of this rank; otherwise by `feeders` threads.  Returns / posts {'errors', 'stats'}."""
    from . import comm as dmcomm, signal as dmsignal
    communicator = rdv = None
    if world > 1:       # collectively, before any work: a rank that cannot join fails the run at once, not after its share of the reads
        rdv = dmcomm.FileRendezvous(os.path.join(moptions['outFolder'], '.rendezvous'), rank, world)
    try:
        return _stream_rank_body(moptions, rank, world, device, work, result_q, feeders, feeder_procs, rdv)
    except BaseException as exc:
        if rdv is not None:
            rdv.abort('%s: %s' % (type(exc).__name__, exc))      # the other ranks stop waiting for this one within milliseconds
        raise


def _stream_rank_body(moptions, rank, world, device, work, result_q, feeders, feeder_procs, rdv):
    from . import comm as dmcomm, signal as dmsignal
    communicator = None
    if world > 1:
        communicator = dmcomm.Communicator.from_rendezvous(device, rdv)
        if rank == 0 and moptions.get('outLevel', 2) <= 1:      # --outLevel 0 / 1: which collective library merges the counters of this run
            path, version = dmcomm.rccl_info()
            print('RCCL: %d ranks over %s (ncclGetVersion %d%s)' % (world, path, version, '; DEEPMOD_RCCL_LIBRARY' if os.environ.get('DEEPMOD_RCCL_LIBRARY') else ''), flush=True)
    use_procs = feeder_procs > 0 and hasattr(work, 'get')
    backend = None if use_procs else HipBackend(moptions, device)       # with feeder processes: created once they are running
    eng = StreamEngine(moptions, backend, rank, world)
    if moptions.get('_t_manager'):
        eng.stats['at_rank_start'] = time.time() - moptions['_t_manager']     # process start-up + imports of this rank
    if moptions.get('Ref') and os.path.isfile(moptions['Ref']):
        from . import readmap
        eng.set_reference_lengths({c: len(s) for c, s in readmap.read_fasta(moptions['Ref']).items()})
    if use_procs:
        eng.run_processes(work, feeder_procs, device, make_backend=lambda: HipBackend(moptions, device))
        backend = eng.backend
    else:
        batches = _drain(work) if hasattr(work, 'get') else iter(work)
        eng.run(batches, feeders=feeders, make_normalizer=lambda: dmsignal.SignalNormalizer(device))
    if communicator is not None:
        rounds = iter(range(1 << 30))
        gather = lambda obj: rdv.all_gather_json('finalize_%d' % next(rounds), obj)
        scatter_fn = lambda s: s.reduce_scatter(communicator)
    else:
        gather = scatter_fn = None
    eng.finalize(gather, scatter_fn)
    if moptions.get('_t_manager'):
        eng.stats['at_bed_written'] = time.time() - moptions['_t_manager']
    if communicator is not None:
        eng.stats.update({'comm_' + k: v for k, v in communicator.stats().items()})
        communicator.close()
    backend.close()
    if moptions.get('_t_manager'):
        eng.stats['at_rank_end'] = time.time() - moptions['_t_manager']
    out = {'rank': rank, 'errors': dict(eng.errors), 'stats': dict(eng.stats)}
    if result_q is not None:
        result_q.put(out)
    return out
