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

import os
import queue
import threading
import time
from collections import defaultdict, deque
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from . import predstore, rawreads

PAD = 100          # zero-padded feature rows on both sides of a read (myDetect.py:850-851)
HALF = 10          # windowsize // 2


class Prepared:
    """One worker batch, ready for the device.  Rows of the reads are concatenated, reads grouped by (contig, strand)."""
    __slots__ = ('rows', 'pos', 'flags', 'n_rows', 'groups', 'n_windows', 'n_reads', 'errors', 'contig_len', 'timing', 'files')

    def __init__(self):
        self.rows = np.zeros((0, 7), np.float32)
        self.pos = np.zeros(0, np.int64)          # [n_rows classified rows | extra rows]
        self.flags = np.zeros(0, np.uint8)
        self.n_rows = 0
        self.groups: List[Tuple[str, str, int, int, int, int]] = []   # (chr, strand, row_lo, row_hi, extra_lo, extra_hi)
        self.n_windows = 0
        self.n_reads = 0
        self.errors: Dict[str, List[str]] = defaultdict(list)
        self.contig_len: Dict[str, int] = {}
        self.timing: Dict[str, float] = defaultdict(float)
        self.files: List[str] = []


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
    n_al = (eo[1:] - eo[:-1]) - start_clip - end_clip                     # aligned events per read
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
        if n_al[i] < 50:
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
            refb.append(bmi['refbase'].astype('S1'))
            readb.append(bmi['readbase'].astype('S1'))
            refi.append(bmi['refbasei'].astype(np.int64))
        evb.append(rawreads.event_bases(rd['events']['model_state']).astype('S1'))
        metas.append(rd)
    off = lambda parts: np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    pk = {'tx': np.concatenate(tx), 'refbase': np.concatenate(refb), 'readbase': np.concatenate(readb),
          'refbasei': np.concatenate(refi), 'evbase': np.concatenate(evb), 'row_off': off(tx), 'bmi_off': off(refb),
          'ev_off': off(evb), 'reads': metas, 'contig_len': {}}
    rows_from_packed(pk, base, src, out)


def finish(out: Prepared) -> Prepared:
    """Group the collected reads by (contig, strand) and build the contiguous host arrays of the batch."""
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
    if rows:
        out.rows = np.ascontiguousarray(np.concatenate(rows), np.float32)
        out.pos = np.concatenate(pos + xpos)
        out.flags = np.concatenate(flags + xflags)
    out.n_rows = r
    out._pieces = []
    return out


class _PreparedBuilder(Prepared):
    __slots__ = ('_pieces',)

    def __init__(self):
        super().__init__()
        self._pieces = []


def prepare_batch(moptions, files: List[str], make_normalizer=None) -> Prepared:
    """Host side of one worker batch (the reference's mDetect1 up to the call of mPredict1, myDetect.py:392-465, :488-715):
    raw containers go through signal normalisation, alignment records, dm_map_read and get_Feature; feature containers
    enter at the prediction step."""
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
    finish(out)
    out.timing['rows'] += time.perf_counter() - t0
    return out


# ---------------------------------------------------------------------------------------------
# device backend (C ABI) and the rank-local engine
# ---------------------------------------------------------------------------------------------
class HipBackend:
    """The device side of the engine through libdeepmod_hip.so: one model, one in-order stream, growable staging
    buffers, one PositionSummary per contig x strand."""

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
        self._dm = dm
        self._buf = {}            # name -> DeviceArray (grow-only)
        self._inflight = deque()  # host batches whose uploads may still be in flight

    def _device_buffer(self, name, nbytes):
        cur = self._buf.get(name)
        if cur is None or cur.nbytes < nbytes:
            if cur is not None:
                self.model.sync()
                cur.free()
            cur = self._dm.DeviceArray((int(nbytes * 1.25) + 4096,), np.uint8, self.device)
            self._buf[name] = cur
        return cur

    def new_summary(self, length: int):
        from . import summary
        s = summary.PositionSummary(length, self.device)
        s.follow(self.model)
        return s

    def submit(self, pb: Prepared, summaries) -> None:
        """Queue one batch: upload, classify every row (windows assembled on the device), accumulate per group.
        Returns after enqueue; the host arrays stay referenced until the stream has passed them."""
        if pb.n_rows == 0:
            return
        R, T = pb.n_rows, len(pb.pos)
        d_rows = self._device_buffer('rows', R * 28)
        d_pos = self._device_buffer('pos', T * 8)
        d_flags = self._device_buffer('flags', T)
        d_cls = self._device_buffer('cls', R)
        self.model.upload_async(d_rows.ptr, pb.rows)
        self.model.upload_async(d_pos.ptr, pb.pos)
        self.model.upload_async(d_flags.ptr, pb.flags)
        # window centred on row r -> cls[r]; rows 0..9 and R-10..R-1 are padding of the first / last read
        self.model.predict_rows_device(d_rows.ptr, R, HALF, R - 2 * HALF, d_cls.ptr + HALF)
        for (c, s, lo, hi, xlo, xhi) in pb.groups:
            summ = summaries(c, s, pb.contig_len.get(c, 0))
            summ.add_classified_device(d_pos.ptr + 8 * lo, d_flags.ptr + lo, d_cls.ptr + lo, hi - lo)
            if xhi > xlo:
                summ.add_device(d_pos.ptr + 8 * (R + xlo), d_flags.ptr + R + xlo, xhi - xlo)
        self._inflight.append(pb)
        if len(self._inflight) > 2:      # bound the host memory held for uploads: drain, then drop the oldest batches
            self.model.sync()
            self._inflight.clear()

    def sync(self):
        self.model.sync()
        self._inflight.clear()

    def close(self):
        self.sync()
        for b in self._buf.values():
            b.free()
        self._buf = {}
        self.sess.close()


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

    def finalize(self, gather, reduce_fn, write: bool = True) -> Dict[Tuple[str, str], bytes]:
        """Merge over ranks and write the BED files on rank 0.
        gather(obj) -> list of every rank's obj (control plane); reduce_fn(summary) sums one summary over the ranks
        into rank 0 (data plane).  All ranks walk the union of contig x strand keys in the same order."""
        from . import summary as dmsum
        t0 = time.perf_counter()
        mine = {"%s\t%s" % k: self.summaries[k].length for k in self.summaries}
        lens = dict(self.ref_len)
        everyone = gather({"keys": mine, "len": lens}) if self.world > 1 else [{"keys": mine, "len": lens}]
        keys = sorted(set(k for e in everyone for k in e["keys"]))
        beds = {}
        for key in keys:
            chrom, strand = key.split("\t")
            length = max([e["keys"].get(key, 0) for e in everyone] + [e["len"].get(chrom, 0) for e in everyone])
            s = self.summaries.get((chrom, strand))
            if s is None:                               # this rank saw no read of that contig x strand: zeros
                s = self.summaries[(chrom, strand)] = self.backend.new_summary(length)
            s.grow(length)                              # exactly the common length (no growth slack): equal counts on every rank
            if self.world > 1:
                reduce_fn(s)
            if self.rank == 0:
                touch, cov, mod = s.fetch()
                bed = dmsum.bed_lines(chrom, strand, self.mo['Base'], touch, cov, mod)
                beds[(chrom, strand)] = bed
                if write and len(bed) > 0:          # the reference writes no file for an empty table (myDetect.py:1109)
                    with open('%s/mod_pos.%s%s.%s.bed' % (self.mo['outFolder'], chrom, strand, self.mo['Base']), 'wb') as fh:
                        fh.write(bed)
            s.close()
        self.summaries = {}
        self.stats['merge+bed'] += time.perf_counter() - t0
        return beds


# ---------------------------------------------------------------------------------------------
# process entry: one rank = one GPU
# ---------------------------------------------------------------------------------------------
def _drain(q):
    while True:
        try:
            item = q.get(block=False)
        except Exception:
            return
        yield item[0]


def stream_rank_main(moptions, rank: int, world: int, device: int, work, result_q=None, feeders: int = 2):
    """Body of one GPU process.  `work`: a shared queue of (files, subfolder, batchid) items drained by all ranks,
    or a list of file lists (static shard).  Returns / posts {'errors', 'stats'}."""
    from . import comm as dmcomm, signal as dmsignal
    communicator = rdv = None
    if world > 1:       # collectively, before any work: a rank that cannot join fails the run at once, not after its share of the reads
        rdv = dmcomm.FileRendezvous(os.path.join(moptions['outFolder'], '.rendezvous'), rank, world)
        communicator = dmcomm.Communicator.from_rendezvous(device, rdv)
    backend = HipBackend(moptions, device)
    eng = StreamEngine(moptions, backend, rank, world)
    if moptions.get('Ref') and os.path.isfile(moptions['Ref']):
        from . import readmap
        eng.set_reference_lengths({c: len(s) for c, s in readmap.read_fasta(moptions['Ref']).items()})
    batches = _drain(work) if hasattr(work, 'get') else iter(work)
    eng.run(batches, feeders=feeders, make_normalizer=lambda: dmsignal.SignalNormalizer(device))
    if communicator is not None:
        gather = lambda obj: rdv.all_gather_json('summary_keys', obj)
        reduce_fn = lambda s: s.reduce(communicator, 0)
    else:
        gather = reduce_fn = None
    eng.finalize(gather, reduce_fn)
    if communicator is not None:
        eng.stats.update({'comm_' + k: v for k, v in communicator.stats().items()})
        communicator.close()
    backend.close()
    out = {'rank': rank, 'errors': dict(eng.errors), 'stats': dict(eng.stats)}
    if result_q is not None:
        result_q.put(out)
    return out
