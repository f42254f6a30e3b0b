"""CPU: feeder processes of the streaming detect (deepmod_amd/stream.py: feeder_process_main / prepared_from_shm /
StreamEngine.run_processes).  A batch prepared in a worker process and handed over through a shared-memory file must be the
batch prepare_batch() builds in-process, and the engine fed by processes must produce the BED bytes of the engine fed by
threads (device calls replaced by the oracle stand-in of tests/cpu_backend.py, as in test_stream_gloo.py)."""
import json
import multiprocessing
import os
import queue

import pytest

import numpy as np

from deepmod_amd import stream, synth, synth_reads
from cpu_backend import OracleBackend


def _files(tmp_path):
    return synth_reads.write_synthetic_run(str(tmp_path / 'in'), n_reads=9, reads_per_file=2, genome_len=6000, seed=3, chrom='chrA',
                                           min_len=120, max_len=400)


def test_batch_through_shared_memory_equals_in_process_batch(tmp_path):
    files = _files(tmp_path)
    mo = {'Base': 'C', 'outFolder': str(tmp_path), 'fnum': 7, 'hidden': 100, 'windowsize': 21}
    work, ready = queue.Queue(), queue.Queue()
    work.put((files[:3], 0, 0))
    work.put((files[3:], 0, 1))
    shm = str(tmp_path / 'shm')
    os.makedirs(shm)
    stream.feeder_process_main(mo, work, ready, 0, shm, 0)          # the process body, run here
    metas = []
    while True:
        m = ready.get()
        if m is None:
            break
        assert 'failed' not in m, m
        metas.append(m)
    assert len(metas) == 2
    for m, fl in zip(metas, (files[:3], files[3:])):
        got = stream.prepared_from_shm(m)
        assert not os.path.exists(m['path'])                         # unlinked as soon as it is mapped
        ref = stream.prepare_batch(mo, fl)
        assert got.n_rows == ref.n_rows and got.n_reads == ref.n_reads and got.n_windows == ref.n_windows
        assert got.groups == ref.groups and got.contig_len == ref.contig_len and dict(got.errors) == dict(ref.errors)
        assert np.array_equal(got.rows, ref.rows) and np.array_equal(got.pos, ref.pos) and np.array_equal(got.flags, ref.flags)
    assert os.listdir(shm) == []


def _bed(mo, backend, run):
    eng = stream.StreamEngine(mo, backend)
    run(eng)
    beds = eng.finalize(None, None, write=False)
    return {k: bytes(v) for k, v in beds.items()}, dict(eng.stats)


def test_engine_fed_by_processes_equals_engine_fed_by_threads(tmp_path):
    files = _files(tmp_path)
    w = synth.synthetic_weights(26, 4.0)
    mo = {'Base': 'C', 'outFolder': str(tmp_path / 'out'), 'fnum': 7, 'hidden': 100, 'windowsize': 21, 'feeder_slots': False}
    os.makedirs(mo['outFolder'])
    items = [files[i:i + 2] for i in range(0, len(files), 2)]
    ref, st_ref = _bed(mo, OracleBackend(w), lambda eng: eng.run(iter(items), feeders=2))

    ctx = multiprocessing.get_context('spawn')
    with ctx.Manager() as mgr:
        work = mgr.Queue()
        for i, it in enumerate(items):
            work.put((it, 0, i))
        got, st = _bed(mo, OracleBackend(w), lambda eng: eng.run_processes(work, 2, 0, ctx))
    assert got == ref and len(ref) > 0
    assert st['reads'] == st_ref['reads'] and st['windows'] == st_ref['windows']
    assert not os.path.exists(stream.shm_dir_for(mo))


def test_slot_ring_hand_over(tmp_path):
    files = _files(tmp_path)
    w = synth.synthetic_weights(26, 4.0)
    mo = {'Base': 'C', 'outFolder': str(tmp_path / 'out'), 'fnum': 7, 'hidden': 100, 'windowsize': 21, 'feeder_slot_mb': 1}
    os.makedirs(mo['outFolder'])
    items = [files[i:i + 1] for i in range(len(files))] * 3          # more batches than slots: slots are recycled
    ref, st_ref = _bed(mo, OracleBackend(w), lambda eng: eng.run(iter(items), feeders=2))
    ctx = multiprocessing.get_context('spawn')
    backend = OracleBackend(w)
    with ctx.Manager() as mgr:
        work = mgr.Queue()
        for i, it in enumerate(items):
            work.put((it, 0, i))
        got, st = _bed(mo, backend, lambda eng: eng.run_processes(work, 2, 0, ctx))
    assert got == ref and st['reads'] == st_ref['reads']
    assert st['slot_batches'] == len(items)                       # every batch went through a (recycled) slot
    assert not os.path.exists(stream.shm_dir_for(mo))


def test_remote_signal_normalizer_protocol(tmp_path):
    """The feeder side of the signal server (stream.RemoteSignalNormalizer: request layout, growth of the request file,
    error channel), served here by a thread that answers from the signal oracle instead of the device."""
    import mmap
    import threading
    from oracle import signal_oracle
    from deepmod_amd import _lib, rawreads
    requests, answers = queue.Queue(), queue.Queue()

    def serve():
        while True:
            req = requests.get()
            if req is None:
                return
            wid, path, size, n, n_raw, n_ev = req
            with open(path, 'r+b') as fh:
                mm = mmap.mmap(fh.fileno(), size)
            o = stream._sig_layout(n, n_raw, n_ev)
            raw = np.frombuffer(mm, np.int16, n_raw, o['raw'])
            ro = np.frombuffer(mm, np.int64, n + 1, o['raw_off'])
            eo = np.frombuffer(mm, np.int64, n + 1, o['ev_off'])
            st = np.frombuffer(mm, np.uint64, n_ev, o['ev_start'])
            ln = np.frombuffer(mm, np.uint64, n_ev, o['ev_length'])
            if n_ev and int(ln[0]) == 0:
                answers.put('bad request')
                continue
            for r in range(n):
                ev = np.zeros(eo[r + 1] - eo[r], dtype=rawreads.EVENT_DTYPE)
                ev['start'], ev['length'] = st[eo[r]:eo[r + 1]], ln[eo[r]:eo[r + 1]]
                sig, norm = signal_oracle.mnormalized(raw[ro[r]:ro[r + 1]], ev)
                mean, stdv, fe = signal_oracle.event_stats(sig, ev)
                np.frombuffer(mm, np.float32, n_ev, o['mean'])[eo[r]:eo[r + 1]] = mean
                np.frombuffer(mm, np.float32, n_ev, o['stdv'])[eo[r]:eo[r + 1]] = stdv
                np.frombuffer(mm, np.float64, 6 * n, o['norm6'])[6 * r:6 * r + 6] = [norm[k] for k in ('mshift', 'mscale', 'read_med', 'read_mad', 'lower_lim', 'upper_lim')]
                np.frombuffer(mm, np.int64, n, o['first_empty'])[r] = fe
            answers.put(None)

    th = threading.Thread(target=serve, daemon=True)
    th.start()
    rng = np.random.default_rng(4)
    norm = stream.RemoteSignalNormalizer(0, str(tmp_path), requests, answers)
    for n_reads, n_samples in ((3, 4000), (2, 700_000)):           # the second batch outgrows the first request file
        reads = []
        for _ in range(n_reads):
            raw = rng.integers(300, 900, n_samples).astype(np.int16)
            length = rng.integers(3, 12, n_samples // 10).astype(np.uint64)
            start = np.concatenate([[5], 5 + np.cumsum(length)[:-1]]).astype(np.uint64)
            reads.append((raw, start, length))
        got = norm.event_stats_batch(reads)
        assert len(got) == n_reads
        for (raw, start, length), (mean, stdv, nd, fe) in zip(reads, got):
            ev = np.zeros(len(start), dtype=rawreads.EVENT_DTYPE)
            ev['start'], ev['length'] = start, length
            sig, ref_norm = signal_oracle.mnormalized(raw, ev)
            ref_mean, ref_stdv, ref_fe = signal_oracle.event_stats(sig, ev)
            assert fe == ref_fe and np.array_equal(mean, ref_mean, equal_nan=True) and np.array_equal(stdv, ref_stdv, equal_nan=True)
            assert nd['mshift'] == ref_norm['mshift'] and nd['upper_lim'] == ref_norm['upper_lim']
    one = norm.event_stats(*reads[0])
    assert np.array_equal(one[0], got[0][0], equal_nan=True) and one[4] is None
    bad = (reads[0][0], reads[0][1], np.zeros_like(reads[0][2]))
    try:
        norm.event_stats_batch([bad])
        raise AssertionError('the server error must surface')
    except _lib.DeepModHipError as exc:
        assert 'bad request' in str(exc)
    requests.put(None)
    th.join(timeout=5)
    norm.close()


class _OracleNormalizer:
    """TEST stand-in for the device signal stage (deepmod_amd.signal.SignalNormalizer): both call forms of the interface, served
    from oracle/signal_oracle.py, so that the raw path of prepare_batch runs on a CPU box."""

    def _one(self, raw, st, ln):
        from oracle import signal_oracle
        from deepmod_amd import rawreads
        ev = np.zeros(len(st), dtype=rawreads.EVENT_DTYPE)
        ev['start'], ev['length'] = st, ln
        sig, norm = signal_oracle.mnormalized(raw, ev)
        mean, stdv, fe = signal_oracle.event_stats(sig, ev)
        return mean, stdv, norm, fe

    def event_stats_batch(self, reads):
        return [self._one(*r) for r in reads]

    def event_stats_arrays(self, raw_parts, raw_off, ev_start, ev_length, ev_off):
        raw = np.concatenate(raw_parts)
        n = len(raw_off) - 1
        mean, stdv, fe = np.empty(len(ev_start), np.float32), np.empty(len(ev_start), np.float32), np.empty(n, np.int64)
        for r in range(n):
            m, s, _, f = self._one(raw[raw_off[r]:raw_off[r + 1]], ev_start[ev_off[r]:ev_off[r + 1]], ev_length[ev_off[r]:ev_off[r + 1]])
            mean[ev_off[r]:ev_off[r + 1]], stdv[ev_off[r]:ev_off[r + 1]], fe[r] = m, s, f
        return mean, stdv, fe


def _same_batch(a, b):
    assert a.n_rows == b.n_rows and a.n_reads == b.n_reads and a.n_windows == b.n_windows
    assert a.groups == b.groups, (a.groups, b.groups)
    used = {g[0] for g in a.groups}          # (a contig that no surviving read names never gets counters: its length is not compared)
    assert {c: a.contig_len[c] for c in used} == {c: b.contig_len[c] for c in used}
    assert np.array_equal(a.rows, b.rows) and np.array_equal(a.pos, b.pos) and np.array_equal(a.flags, b.flags)
    assert a.f32 == b.f32
    norm = lambda e: {k.split(':')[0]: sorted(v) for k, v in e.items() if v}
    assert norm(a.errors) == norm(b.errors), (dict(a.errors), dict(b.errors))


def _compact_form_is_the_classic_form_without_the_other_bases(c, classic):
    """dm_rows_emit's compact form: the same feature rows; sel = the rows that are centres of windows on a base of interest (flag
    bit 0), their positions and flags, then the same extras - rows of other bases appear nowhere (the summary ignores them)."""
    R = classic.n_rows
    assert c.sel is not None and classic.sel is None and c.n_rows == R and np.array_equal(c.rows, classic.rows)
    want = np.flatnonzero((classic.flags[:R] & 1) != 0)
    assert np.array_equal(c.sel, want) and len(want) < 0.5 * R
    S = len(want)
    assert np.array_equal(c.pos[:S], classic.pos[want]) and (c.flags[:S] == 3).all() and (classic.flags[want] == 3).all()
    assert np.array_equal(c.pos[S:], classic.pos[R:]) and np.array_equal(c.flags[S:], classic.flags[R:])
    assert len(c.groups) == len(classic.groups)
    for g, h in zip(c.groups, classic.groups):
        assert g[:6] == h[:6]
        assert np.array_equal(c.sel[g[6]:g[7]], want[(want >= g[2]) & (want < g[3])])
    assert c.n_windows == classic.n_windows and c.n_reads == classic.n_reads and c.contig_len == classic.contig_len


def test_compiled_batch_equals_python_batch_on_feature_and_packed_containers(tmp_path):
    """dm_rows_add_packed + dm_rows_info + dm_rows_emit against the per-read numpy path (stream._prepare_batch_py): feature
    containers of both formats, two contigs, both strands, a read too short to be called, deletions (extra rows)."""
    from deepmod_amd import predstore
    a = synth_reads.write_synthetic_run(str(tmp_path / 'a'), n_reads=9, reads_per_file=3, genome_len=6000, seed=3, chrom='chrB',
                                        min_len=120, max_len=400)
    b = synth_reads.write_synthetic_run(str(tmp_path / 'b'), n_reads=6, reads_per_file=3, genome_len=5000, seed=4, chrom='chrA',
                                        min_len=120, max_len=400)
    packed = []
    for i, f in enumerate(b):
        p = str(tmp_path / 'b' / ('packed_%d%s' % (i, predstore.CONTAINER_SUFFIX)))
        predstore.save_packed_container(p, predstore.load_feature_container(f), {'chrA': 5000})
        os.remove(f)
        packed.append(p)
    short = synth_reads.write_synthetic_run(str(tmp_path / 'c'), n_reads=2, reads_per_file=2, genome_len=3000, seed=5, chrom='chrB',
                                            min_len=30, max_len=40)
    files = a + packed + short
    for base in 'CA':
        mo = {'Base': base, 'outFolder': str(tmp_path), 'fnum': 7, 'hidden': 100, 'windowsize': 21, 'select_base': False}
        got = stream._prepare_batch_c(mo, files)
        ref = stream._prepare_batch_py(mo, files)
        _same_batch(got, ref)
        _compact_form_is_the_classic_form_without_the_other_bases(stream._prepare_batch_c(dict(mo, select_base=True), files), got)
        assert got.n_reads == 15 and len(got.errors['Less Event']) == 2 and len(got.groups) >= 3
        assert (got.flags[got.n_rows:] & 1).all() and len(got.pos) > got.n_rows        # extras exist and are all of the wanted base


def test_compiled_batch_equals_python_batch_on_raw_containers(tmp_path, monkeypatch):
    """dm_events_merge + dm_rows_add_raw (alignment walk, get_Feature rows) against rawreads.getEvent / readmap.map_records /
    features.get_Feature / stream.rows_from_reads, the signal stage served by the oracle on both sides."""
    files, fasta = synth_reads.write_synthetic_raw_run(str(tmp_path / 'in'), n_reads=14, reads_per_file=5, genome_len=30000, seed=6,
                                                       chrom='chrS', min_len=300, max_len=1200)
    mo = {'Base': 'C', 'outFolder': str(tmp_path), 'fnum': 7, 'hidden': 100, 'windowsize': 21, 'Ref': fasta, 'alignStr': 'minimap2',
          'region': [[None, None, None]], 'ConUnk': True, 'SignalGroup': 'simple', 'outLevel': 3}
    norm = _OracleNormalizer()
    mo['select_base'] = False
    got = stream._prepare_batch_c(dict(mo), files, lambda: norm)
    ref = stream._prepare_batch_py(dict(mo), files, lambda: norm)
    _same_batch(got, ref)
    dev = stream._prepare_batch_c(dict(mo, select_base=True), files, lambda: norm)
    _compact_form_is_the_classic_form_without_the_other_bases(dev, got)
    assert got.n_reads >= 12 and got.n_windows > 5000 and got.rows[:, :4].sum() > 0 and got.rows[:, 4:].any()
    # round 5: a compact batch of raw reads is handed over in the DEVICE form - (mean, stdv, length) per event, a class byte per row, a
    # descriptor per read - and dm_rows_assemble (here: its host restatement stream.assemble_rows, behind Prepared.rows) rebuilds the
    # same matrix bit for bit; rows_on_device = False keeps the rows on the host
    assert dev._rows is None and dev.ev3 is not None and dev.code.shape == (dev.n_rows,) and dev.rdesc.shape == (dev.n_reads, 4)
    assert dev.ev3.nbytes + dev.code.nbytes + dev.rdesc.nbytes < 0.5 * got.rows.nbytes
    assert (np.diff(dev.rdesc[:, 0]) > 0).all() and dev.rdesc[0, 0] == 0 and (dev.rdesc[:, 3] >= dev.rdesc[:, 2]).all()
    host = stream._prepare_batch_c(dict(mo, select_base=True, rows_on_device=False), files, lambda: norm)
    assert host.ev3 is None and np.array_equal(host._rows, got.rows) and np.array_equal(host.sel, dev.sel) and np.array_equal(host.pos, dev.pos)
    assert host.f32 == dev.f32 == got.f32
    # ... and through a feeder process' shared-memory hand-over (the process body run here, its signal stage served by the oracle)
    from deepmod_amd import signal as dmsignal
    monkeypatch.setattr(dmsignal, 'SignalNormalizer', lambda device: norm)
    work, ready = queue.Queue(), queue.Queue()
    work.put((files, 0, 0))
    shm = str(tmp_path / 'shm_dev')
    os.makedirs(shm)
    stream.feeder_process_main(dict(mo, select_base=True), work, ready, 0, shm, 0)
    meta = ready.get()
    assert meta is not None and 'failed' not in meta, meta
    assert meta['dev'] == (len(dev.ev3), dev.n_reads)
    via = stream.prepared_from_shm(meta)
    assert via._rows is None and np.array_equal(via.ev3, dev.ev3) and np.array_equal(via.code, dev.code) and np.array_equal(via.rdesc, dev.rdesc)
    assert np.array_equal(via.rows, got.rows) and np.array_equal(via.sel, dev.sel) and np.array_equal(via.pos, dev.pos) and via.groups == dev.groups
    assert ready.get() is None and os.listdir(shm) == []
    # a region filter that keeps the first half of the contig only, and a region on another contig (nothing passes)
    for region, expect_some in (([['chrS', None, 15000]], True), ([['chrT', None, None]], False)):
        g2 = stream._prepare_batch_c(dict(mo, region=region), files, lambda: norm)
        r2 = stream._prepare_batch_py(dict(mo, region=region), files, lambda: norm)
        _same_batch(g2, r2)
        assert (g2.n_reads > 0) == expect_some and g2.n_reads < got.n_reads


class _PostingOracleNormalizer(_OracleNormalizer):
    """TEST stand-in for the resident form of the signal stage: post_arrays() records the request and builds, from the signal oracle, the block
    dm_signal_event_stats_device would leave on the device - ev3 [n_events][3] = (mean, stdv, length) of every merged event of the batch, the
    fall-back values merged in for events at or behind a read's first empty event."""

    def __init__(self):
        self.blocks = {}

    def post_arrays(self, raw_parts, raw_off, ev_start, ev_length, ev_off, first_empty, fb_mean=None, fb_stdv=None):
        mean, stdv, fe = self.event_stats_arrays(raw_parts, raw_off, ev_start, ev_length, ev_off)
        assert np.array_equal(fe, first_empty), 'dm_signal_plan_batch and the signal stage disagree on the first empty event'
        ev3 = np.empty((len(ev_start), 3), np.float32)
        ev3[:, 0], ev3[:, 1], ev3[:, 2] = mean, stdv, ev_length.astype(np.float64).astype(np.float32)
        for r in range(len(raw_off) - 1):
            lo = ev_off[r] + min(int(first_empty[r]), int(ev_off[r + 1] - ev_off[r]))
            if lo < ev_off[r + 1]:
                assert fb_mean is not None, 'a read with an empty event and no fall-back values'
                ev3[lo:ev_off[r + 1], 0], ev3[lo:ev_off[r + 1], 1] = fb_mean[lo:ev_off[r + 1]], fb_stdv[lo:ev_off[r + 1]]
        key = (0, len(self.blocks) + 1)
        self.blocks[key] = ev3
        return key


def test_resident_form_of_a_raw_batch_equals_the_device_form(tmp_path):
    """Round 6: with a signal stage that keeps its statistics on the device (post_arrays), a batch of raw containers hands over NO per-event values -
    only a class byte per row, a descriptor per read whose event indices point into the signal stage's block, the window list, positions and flags.
    Assembled from that block (stream.assemble_rows = the host restatement of dm_rows_assemble) the feature rows are those of the device form, and
    everything else of the batch is identical.  One read gets an empty event (its table is cut there, the fall-back values take over)."""
    from deepmod_amd import npzmap
    files, fasta = synth_reads.write_synthetic_raw_run(str(tmp_path / 'in'), n_reads=14, reads_per_file=5, genome_len=30000, seed=6,
                                                       chrom='chrS', min_len=300, max_len=1200)
    # an event that starts behind the end of the signal: an EMPTY slice at index 60 of the second read of the first container (<= 500: the table keeps its length,
    # events from there on keep the basecaller's values, myDetect.py:334-340)
    z = {k: np.array(v) for k, v in npzmap.load(files[0]).items()}          # copies: the file is rewritten below, its mapping must not be read again
    eo = z['ev_off']
    st = np.array(z['ev_start'])
    st[eo[1] + 60] = int(z['raw_off'][2] - z['raw_off'][1]) + 1000
    z['ev_start'] = st
    npzmap.savez_aligned(files[0], **z)
    mo = {'Base': 'C', 'outFolder': str(tmp_path), 'fnum': 7, 'hidden': 100, 'windowsize': 21, 'Ref': fasta, 'alignStr': 'minimap2',
          'region': [[None, None, None]], 'ConUnk': True, 'SignalGroup': 'simple', 'outLevel': 3, 'select_base': True}
    plain = _OracleNormalizer()
    dev = stream._prepare_batch_c(dict(mo), files, lambda: plain)
    assert dev.ev3 is not None and dev.sig is None
    posting = _PostingOracleNormalizer()
    res = stream._prepare_batch_c(dict(mo), files, lambda: posting)
    assert res.sig == (0, 1) and res.ev3 is None and res._rows is None and res.code is not None
    block = posting.blocks[res.sig]
    assert np.array_equal(res.code, dev.code) and np.array_equal(res.sel, dev.sel) and np.array_equal(res.pos, dev.pos) and np.array_equal(res.flags, dev.flags)
    assert res.groups == dev.groups and res.n_rows == dev.n_rows and res.n_reads == dev.n_reads and res.f32 == dev.f32 and res.contig_len == dev.contig_len
    assert np.array_equal(res.rdesc[:, 0], dev.rdesc[:, 0]) and (res.rdesc[:, 3] - res.rdesc[:, 2] == dev.rdesc[:, 3] - dev.rdesc[:, 2]).all()
    rows = stream.assemble_rows(block, res.code, res.rdesc, res.n_rows)
    assert np.array_equal(rows, dev.rows, equal_nan=True) and rows[:, 4:].any()
    # the block holds every merged event of the batch in the order of the request, the batch's own ev3 the ones its rows cover in the order of its groups
    assert len(block) >= len(dev.ev3) and res.rdesc[:, 3].max() <= len(block) and not np.array_equal(block[:len(dev.ev3)], dev.ev3)
    # stats_on_device = False: the same normalizer is asked for host arrays and the batch comes in the device form
    host = stream._prepare_batch_c(dict(mo, stats_on_device=False), files, lambda: posting)
    assert host.sig is None and np.array_equal(host.ev3, dev.ev3, equal_nan=True)
    # through a feeder process' hand-over: nothing per event in the slot
    work, ready = queue.Queue(), queue.Queue()
    work.put((files, 0, 0))
    shm = str(tmp_path / 'shm_res')
    os.makedirs(shm)
    from deepmod_amd import signal as dmsignal
    import pytest
    mp = pytest.MonkeyPatch()
    try:
        mp.setattr(dmsignal, 'SignalNormalizer', lambda device: posting)
        stream.feeder_process_main(dict(mo), work, ready, 0, shm, 0)
    finally:
        mp.undo()
    meta = ready.get()
    assert meta is not None and 'failed' not in meta, meta
    assert tuple(meta["sig"]) == (0, 2) and meta["dev"] == (0, res.n_reads)
    via = stream.prepared_from_shm(meta)
    assert via.sig == (0, 2) and via.ev3 is None and np.array_equal(via.code, res.code) and np.array_equal(via.rdesc, res.rdesc)
    assert np.array_equal(via.sel, res.sel) and np.array_equal(via.pos, res.pos) and via.groups == res.groups
    try:
        via.rows
        raise AssertionError('a resident batch has no feature rows on the host')
    except ValueError:
        pass
    assert ready.get() is None


def test_posted_signal_requests_protocol(tmp_path):
    """Feeder side of the resident form (stream.RemoteSignalNormalizer.post_arrays): a request is written into one of two alternating request files and
    queued WITHOUT waiting for statistics; request k may reuse the file of request k - 2 only after the server acknowledged having copied it; the request
    layout (stream._sig_layout_res) carries first_empty and, only when needed, the fall-back values; a server-side failure surfaces at the next post.
    GPU process side: stream.SignalResults hands a result to whoever asks for it, before or after it arrives."""
    import mmap
    import threading
    import time
    from deepmod_amd import _lib
    requests, answers = queue.Queue(), queue.Queue()
    norm = stream.RemoteSignalNormalizer(3, str(tmp_path), requests, answers)
    rng = np.random.default_rng(2)

    def batch(n_reads, n_samples, with_fb):
        raws = [rng.integers(300, 900, n_samples).astype(np.int16) for _ in range(n_reads)]
        lens = [rng.integers(3, 12, n_samples // 10).astype(np.uint64) for _ in range(n_reads)]
        starts = [np.concatenate([[5], 5 + np.cumsum(ln)[:-1]]).astype(np.uint64) for ln in lens]
        raw_off = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int64)
        ev_off = np.concatenate([[0], np.cumsum([len(x) for x in lens])]).astype(np.int64)
        st, ln = np.concatenate(starts), np.concatenate(lens)
        fe = np.array([len(x) for x in lens], np.int64)
        fb = (rng.random(len(st)).astype(np.float32), rng.random(len(st)).astype(np.float32)) if with_fb else (None, None)
        return raws, raw_off, st, ln, ev_off, fe, fb

    def read_request(req):
        kind, wid, path, size, n, n_raw, n_ev, seq, with_fb = req
        assert kind == 'res' and wid == 3
        with open(path, 'r+b') as fh:
            mm = mmap.mmap(fh.fileno(), size)
        o = stream._sig_layout_res(n, n_raw, n_ev, with_fb)
        out = {k: np.frombuffer(mm, dt, cnt, o[k]).copy() for k, dt, cnt in (('raw', np.int16, n_raw), ('raw_off', np.int64, n + 1), ('ev_off', np.int64, n + 1),
                                                                                ('ev_start', np.uint64, n_ev), ('ev_length', np.uint64, n_ev), ('first_empty', np.int64, n))}
        if with_fb:
            out['fb_mean'] = np.frombuffer(mm, np.float32, n_ev, o['fb_mean']).copy()
        return seq, path, out

    b1, b2, b3 = batch(3, 4000, False), batch(2, 9000, True), batch(2, 700_000, False)
    k1 = norm.post_arrays(b1[0], b1[1], b1[2], b1[3], b1[4], b1[5])
    k2 = norm.post_arrays(b2[0], b2[1], b2[2], b2[3], b2[4], b2[5], *b2[6])
    assert k1 == (3, 1) and k2 == (3, 2) and requests.qsize() == 2          # both queued, nobody waited
    s1, p1, r1 = read_request(requests.get())
    s2, p2, r2 = read_request(requests.get())
    assert (s1, s2) == (1, 2) and p1 != p2
    assert np.array_equal(r1['raw'], np.concatenate(b1[0])) and np.array_equal(r1['ev_start'], b1[2]) and np.array_equal(r1['first_empty'], b1[5])
    assert np.array_equal(r2['ev_length'], b2[3]) and np.array_equal(r2['fb_mean'], b2[6][0])
    # request 3 wants the file of request 1: it blocks until that one is acknowledged
    done = []
    th = threading.Thread(target=lambda: done.append(norm.post_arrays(b3[0], b3[1], b3[2], b3[3], b3[4], b3[5])), daemon=True)
    th.start()
    time.sleep(0.3)
    assert not done and requests.empty()
    answers.put(('ack', 1, None))
    th.join(timeout=10)
    assert done == [(3, 3)]
    s3, p3, r3 = read_request(requests.get())
    assert s3 == 3 and p3 == p1 and np.array_equal(r3['raw'], np.concatenate(b3[0]))      # the first file again (grown: the batch outgrew it)
    # a failed acknowledgement (the server could not even copy the request) surfaces at the next post
    answers.put(('ack', 2, 'signal server: boom'))
    try:
        norm.post_arrays(b1[0], b1[1], b1[2], b1[3], b1[4], b1[5])
        raise AssertionError('the server error must surface')
    except _lib.DeepModHipError as exc:
        assert 'boom' in str(exc)
    norm.close()
    # the GPU process' registry
    res = stream.SignalResults()
    res.put((3, 1), 'block-a', 1, None)
    assert res.take((3, 1)) == ('block-a', 1, None) and res.pending() == []
    got = []
    th = threading.Thread(target=lambda: got.append(res.take((3, 2), timeout=10)), daemon=True)
    th.start()
    time.sleep(0.1)
    res.put((3, 2), None, 0, 'signal stage: failed')
    th.join(timeout=10)
    assert got == [(None, 0, 'signal stage: failed')]
    try:
        res.take((9, 9), timeout=0.2)
        raise AssertionError('a result that never arrives must time out')
    except RuntimeError as exc:
        assert 'never answered' in str(exc)


def test_compact_batch_without_any_base_of_interest(tmp_path):
    """A batch whose reads hold no base of interest at all (an all-T genome, --Base C): feature rows but no window to classify, no
    positions, no extras - through the compiled path, the shared-memory hand-over and the engine (no BED file is written)."""
    from deepmod_amd import predstore
    files = synth_reads.write_synthetic_run(str(tmp_path / 'in'), n_reads=4, reads_per_file=2, genome_len=4000, seed=3, chrom='chrA',
                                            min_len=120, max_len=300)
    for f in files:                                   # rewrite the tables: every reference base a 'T'
        reads = predstore.load_feature_container(f)
        for rd in reads:
            bmi = rd['base_map_info']
            bmi['refbase'] = np.where(bmi['refbase'] == '-', '-', 'T')
        predstore.save_feature_container(f, reads)
    mo = {'Base': 'C', 'outFolder': str(tmp_path / 'out'), 'fnum': 7, 'hidden': 100, 'windowsize': 21}
    os.makedirs(mo['outFolder'])
    pb = stream._prepare_batch_c(mo, files)
    assert pb.n_rows > 0 and pb.sel is not None and len(pb.sel) == 0 and len(pb.pos) == 0 and pb.n_reads == 4
    work, ready = queue.Queue(), queue.Queue()
    work.put((files, 0, 0))
    shm = str(tmp_path / 'shm')
    os.makedirs(shm)
    stream.feeder_process_main(mo, work, ready, 0, shm, 0)
    meta = ready.get()
    got = stream.prepared_from_shm(meta)
    assert got.n_rows == pb.n_rows and got.sel is not None and len(got.sel) == 0 and np.array_equal(got.rows, pb.rows)
    eng = stream.StreamEngine(mo, OracleBackend(synth.synthetic_weights(26, 4.0)))
    eng.consume(got)
    beds = eng.finalize(None, None)
    assert all(len(b) == 0 for b in beds.values()) and not [f for f in os.listdir(mo['outFolder']) if f.endswith('.bed')]


def _take_all(work, out_q, wid):
    import queue as _q
    got = []
    while True:
        try:
            got.append(work.get(block=False))
        except _q.Empty:
            break
    out_q.put((wid, got))


def test_work_list_hands_every_item_out_once_across_processes():
    """stream.WorkList (the streaming run's h5files_Q without a manager process): spawned processes draining it together get
    every item exactly once, and an exhausted list answers queue.Empty like the reference's `get(block=False)`."""
    import multiprocessing
    import queue as _q
    import pytest
    from deepmod_amd import stream
    ctx = multiprocessing.get_context('spawn')
    items = [(['f%d_%d' % (i, j) for j in range(3)], i // 100, i) for i in range(500)]
    work = stream.WorkList(items, ctx)
    out_q = ctx.Queue()
    procs = [ctx.Process(target=_take_all, args=(work, out_q, w)) for w in range(3)]
    for p in procs:
        p.start()
    got = [out_q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join()
    flat = sorted((it for _, g in got for it in g), key=lambda it: it[2])
    assert flat == items
    assert work.empty()
    with pytest.raises(_q.Empty):
        work.get(block=False)
    assert os.path.exists(work.path)
    work.close()
    assert not os.path.exists(work.path)


def test_damaged_feature_container_is_reported_and_the_batch_goes_on(tmp_path):
    """A packed feature container whose offset tables do not fit their columns (a truncated or damaged file): dm_rows_add_packed
    refuses it as a whole (DM_EINVAL, checked against the column sizes the caller passes), the file goes to the error ledger and the
    other containers of the batch are processed as if it were not there."""
    from deepmod_amd import npzmap, predstore
    files = synth_reads.write_synthetic_run(str(tmp_path / 'in'), n_reads=9, reads_per_file=3, genome_len=8000, seed=3, chrom='chrA',
                                            min_len=200, max_len=600)
    assert len(files) == 3
    mo = {'Base': 'C', 'outFolder': str(tmp_path / 'out'), 'fnum': 7, 'hidden': 100, 'windowsize': 21}
    os.makedirs(mo['outFolder'])
    want = stream._prepare_batch_c(dict(mo), [files[0], files[2]])
    for damage in ('offset past the column', 'decreasing offsets', 'truncated column', 'offset table too short'):
        pk = predstore.load_packed(files[1])
        z = {k: np.array(pk[k]) for k in ('tx', 'refbase', 'readbase', 'refbasei', 'evbase', 'row_off', 'bmi_off', 'ev_off')}
        z['format'] = np.array(2)
        z['meta'] = np.array(json.dumps({'reads': pk['reads'], 'contig_len': {}}))
        if damage == 'offset past the column':
            z['bmi_off'][-1] += 100000
        elif damage == 'decreasing offsets':
            z['row_off'][1], z['row_off'][2] = z['row_off'][2], z['row_off'][1]
        elif damage == 'truncated column':
            z['tx'] = z['tx'][:len(z['tx']) // 2]
        else:
            z['ev_off'] = z['ev_off'][:-1]
        bad = str(tmp_path / 'in' / ('damaged' + predstore.CONTAINER_SUFFIX))
        with open(bad, 'wb') as fh:
            npzmap.savez_aligned(fh, **z)
        got = stream._prepare_batch_c(dict(mo), [files[0], bad, files[2]])
        assert got.errors.get("Cannot open container") == [bad], (damage, dict(got.errors))
        assert got.n_reads == want.n_reads and got.n_rows == want.n_rows and got.groups == want.groups, damage
        assert np.array_equal(got.rows, want.rows) and np.array_equal(got.pos, want.pos) and np.array_equal(got.flags, want.flags), damage
        os.remove(bad)


def test_clips_at_the_ends_of_int64_are_a_ledger_line_not_a_wild_read(tmp_path):
    """Clip values near INT64_MAX / INT64_MIN in a container's metadata: `n_events - start_clip - end_clip` wraps to a
    plausible count, so the clips are checked before any arithmetic with them - the read becomes an index error of the ledger, the other
    reads of the container and the batch go on, nothing is read outside the event column."""
    from deepmod_amd import npzmap, predstore
    files = synth_reads.write_synthetic_run(str(tmp_path / 'in'), n_reads=6, reads_per_file=3, genome_len=8000, seed=5, chrom='chrA',
                                            min_len=200, max_len=600)
    mo = {'Base': 'C', 'outFolder': str(tmp_path / 'out'), 'fnum': 7, 'hidden': 100, 'windowsize': 21}
    os.makedirs(mo['outFolder'])
    want = stream._prepare_batch_c(dict(mo), [files[0]])
    for clips in ((2 ** 63 - 1, 2 ** 63 - 1), (-2 ** 63, -2 ** 63), (2 ** 63 - 1, 0), (0, 2 ** 62), (2 ** 62, 2 ** 62)):
        pk = predstore.load_packed(files[1])
        z = {k: np.array(pk[k]) for k in ('tx', 'refbase', 'readbase', 'refbasei', 'evbase', 'row_off', 'bmi_off', 'ev_off')}
        reads = [dict(r) for r in pk['reads']]
        reads[1]['start_clip'], reads[1]['end_clip'] = clips
        z['format'] = np.array(2)
        z['meta'] = np.array(json.dumps({'reads': reads, 'contig_len': {}}))
        bad = str(tmp_path / 'in' / ('clips' + predstore.CONTAINER_SUFFIX))
        with open(bad, 'wb') as fh:
            npzmap.savez_aligned(fh, **z)
        got = stream._prepare_batch_c(dict(mo), [files[0], bad])
        assert got.n_reads == want.n_reads + 2, (clips, got.n_reads, dict(got.errors))
        assert sum(len(v) for v in got.errors.values()) >= 1, clips
        # the ledger keys are the interface: the compiled and the Python build path file the read under the SAME reason -
        # negative clips an index error, clips merely too large for the read "Less Event" (n_events - clips < 50, myDetect.py:702-705)
        py = stream._prepare_batch_py(dict(mo), [files[0], bad])
        assert {k: sorted(v) for k, v in got.errors.items() if v} == {k: sorted(v) for k, v in py.errors.items() if v}, (clips, dict(got.errors), dict(py.errors))
        key = "Prediction failed: IndexError" if min(clips) < 0 else "Less Event"
        assert got.errors.get(key) == [bad], (clips, dict(got.errors))
        assert py.n_reads == got.n_reads and py.n_rows == got.n_rows
        os.remove(bad)
    # ordinary out-of-range clips (start_clip = 1000 on a read of a few hundred events): "Less Event" on both paths
    pk = predstore.load_packed(files[1])
    z = {k: np.array(pk[k]) for k in ('tx', 'refbase', 'readbase', 'refbasei', 'evbase', 'row_off', 'bmi_off', 'ev_off')}
    reads = [dict(r) for r in pk['reads']]
    reads[1]['start_clip'], reads[1]['end_clip'] = 1000, 0
    z['format'] = np.array(2)
    z['meta'] = np.array(json.dumps({'reads': reads, 'contig_len': {}}))
    bad = str(tmp_path / 'in' / ('clips' + predstore.CONTAINER_SUFFIX))
    with open(bad, 'wb') as fh:
        npzmap.savez_aligned(fh, **z)
    for fn in (stream._prepare_batch_c, stream._prepare_batch_py):
        assert fn(dict(mo), [bad]).errors.get("Less Event") == [bad], fn.__name__


def test_damaged_raw_containers_cost_their_reads_and_nothing_else(tmp_path, capsys):
    """Raw containers with damaged tables through the compiled batch builder, resident and device form: a container that cannot be read or whose offset
    tables do not fit its columns is one line of the error ledger and the batch is the batch of the other containers; an event whose start or length is
    absurd (2^63 and more: numpy's uint64 arithmetic, myDetect.py:272, :334-340) is an empty event or a long one exactly as in the Python restatement;
    a read whose FIRST start is absurd covers no signal: the batched signal call refuses it and the per-read path reports it."""
    import json
    from deepmod_amd import npzmap
    files, fasta = synth_reads.write_synthetic_raw_run(str(tmp_path / 'in'), n_reads=12, reads_per_file=4, genome_len=30000, seed=6,
                                                       chrom='chrS', min_len=300, max_len=1200)
    mo = {'Base': 'C', 'outFolder': str(tmp_path), 'fnum': 7, 'hidden': 100, 'windowsize': 21, 'Ref': fasta, 'alignStr': 'minimap2',
          'region': [[None, None, None]], 'ConUnk': True, 'SignalGroup': 'simple', 'outLevel': 3, 'select_base': True}
    orig = open(files[1], 'rb').read()
    z0 = {k: np.array(v) for k, v in npzmap.load(files[1]).items()}
    e1 = int(z0['ev_off'][1])

    def build(which):
        norm = _PostingOracleNormalizer() if which == 'resident' else _OracleNormalizer()
        fn = stream._prepare_batch_py if which == 'py' else stream._prepare_batch_c
        pb = fn(dict(mo), files, lambda: norm)
        rows = stream.assemble_rows(norm.blocks[pb.sig], pb.code, pb.rdesc, pb.n_rows) if pb.sig is not None else pb.rows
        return pb, rows

    def write(**changed):
        npzmap.savez_aligned(files[1], **dict(z0, **changed))

    def edited(key, index, value):
        a = z0[key].copy()
        a[index] = value
        return {key: a}

    os.rename(files[1], files[1] + '.away')
    without, rows_without = build('device')
    os.rename(files[1] + '.away', files[1])
    assert without.n_reads == 8
    container_level = {
        'truncated file': lambda: open(files[1], 'wb').write(orig[:len(orig) // 2]),
        'not a container': lambda: open(files[1], 'wb').write(b'not a zip at all' * 100),
        'empty file': lambda: open(files[1], 'wb').write(b''),
        'event offsets decrease': lambda: write(**edited('ev_off', 2, e1 - 5)),
        'event offsets past the table': lambda: write(**edited('ev_off', -1, int(z0['ev_off'][-1]) + 1000)),
        'sample offsets past the signal': lambda: write(**edited('raw_off', -1, int(z0['raw_off'][-1]) + 10 ** 6)),
        'sample offsets decrease': lambda: write(**edited('raw_off', 2, int(z0['raw_off'][1]) - 7)),
        'a column shorter than the table': lambda: write(ev_move=z0['ev_move'][:-100]),
        'fewer reads in the meta record than in the tables': lambda: write(meta=np.array(json.dumps(json.loads(str(z0['meta']))[:-1]))),
    }
    for name, damage in container_level.items():
        damage()
        for which in ('resident', 'device'):
            pb, rows = build(which)
            assert dict(pb.errors) == {'Cannot open fast5 or other errors': [files[1]]}, (name, which, dict(pb.errors))
            assert pb.n_reads == 8 and pb.groups == without.groups and np.array_equal(pb.pos, without.pos) and np.array_equal(rows, rows_without), (name, which)
            assert (pb.sig is not None) == (which == 'resident')
    event_level = {
        'an event in the middle starts at 2^64 - 3': edited('ev_start', e1 + 50, 2 ** 64 - 3),
        'an event 2^63 samples long': edited('ev_length', e1 + 50, 2 ** 63),
        'a first start of 2^63 + 11': edited('ev_start', e1, 2 ** 63 + 11),
    }
    for name, change in event_level.items():
        write(**change)
        ref, rows_ref = build('py')                                         # classic form: every row has its position and flags
        classic = stream._prepare_batch_c(dict(mo, select_base=False), files, lambda: _OracleNormalizer())
        assert classic.n_reads == ref.n_reads == 12 and classic.groups == ref.groups and np.array_equal(classic.pos, ref.pos) and np.array_equal(classic.flags, ref.flags)
        assert np.array_equal(classic.rows, rows_ref, equal_nan=True), name
        want = np.flatnonzero((ref.flags[:ref.n_rows] & 1) != 0)
        for which in ('resident', 'device'):
            pb, rows = build(which)
            assert pb.n_reads == 12 and np.array_equal(rows, rows_ref, equal_nan=True), (name, which)
            if which == 'resident' and 'first' in name:
                # dm_signal_plan_batch refuses the batch ("events cover no signal") and it is built read by read, in the classic form
                assert pb.sig is None and pb.sel is None and np.array_equal(pb.pos, ref.pos) and np.array_equal(pb.flags, ref.flags)
                continue
            assert (pb.sig is not None) == (which == 'resident') and np.array_equal(pb.sel, want), (name, which)
            assert np.array_equal(pb.pos[:len(want)], ref.pos[want]) and np.array_equal(pb.pos[len(want):], ref.pos[ref.n_rows:]), (name, which)
    capsys.readouterr()


def test_signal_results_registry():
    """stream.SignalResults: what the signal server threads of a GPU process hand to its batch loop - a result is taken once, by the (feeder, number) of its
    request, in any order; a taker waits for a result that is still being computed and gives up (loudly) on one that never comes."""
    import threading
    import time
    reg = stream.SignalResults()
    reg.put((0, 2), 'block-b', 1)
    reg.put([1, 1], 'block-c', 0, None)
    assert sorted(reg.pending()) == [(0, 2), (1, 1)]
    assert reg.take((1, 1)) == ('block-c', 0, None) and reg.take([0, 2]) == ('block-b', 1, None) and reg.pending() == []
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match='never answered'):
        reg.take((0, 2), timeout=0.2)                         # taken once
    assert 0.15 < time.perf_counter() - t0 < 2.0
    late = threading.Timer(0.2, lambda: reg.put((3, 7), None, 0, 'signal stage: no device'))
    late.start()
    assert reg.take((3, 7), timeout=5.0) == (None, 0, 'signal stage: no device')
    late.join()
