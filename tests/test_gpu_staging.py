"""GPU (-m gpu): the staging entry points of the streaming worker (include/deepmod_hip.h: dm_host_alloc / dm_host_free,
dm_model_h2d_async, dm_model_mark / dm_model_wait_mark): a pipelined sequence of batches through page-locked staging sets gives
the classes of plain synchronous calls, markers order host and device, and a range violation surfaces at the marker."""
import os

import numpy as np
import pytest

from deepmod_amd import _lib, model, stream, synth, synth_reads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("copy", ["dm_model_h2d_async", "dm_model_h2d_ahead"])
def test_pipelined_staging_sets_equal_synchronous_calls(gpu_device, copy):
    w = synth.synthetic_weights(26, 4.0)
    m = model.BiLSTMModel(w, device=gpu_device)
    rng = np.random.default_rng(5)
    batches = [synth.synthetic_windows(n, seed=int(s))[:, 10, :].copy() for n, s in ((5000, 1), (777, 2), (12001, 3), (64, 4), (3000, 5), (9000, 6), (120000, 7), (90, 8),
                                                                                        (100000, 9), (64, 10), (110000, 11), (30, 12))]
    if copy == "dm_model_h2d_ahead":        # the copy stream runs ahead of the queue: many more batches of mixed sizes, sets reused ~30 times each
        sizes = rng.integers(30, 150000, 80)
        pool = synth.synthetic_windows(150000, seed=77)[:, 10, :].copy()
        batches += [pool[int(o):int(o) + int(n)].copy() for n, o in zip(sizes, rng.integers(0, 150000 - sizes))]
    want = [m.predict_read(rows, 10, len(rows) - 20, want_prob=False)[1] for rows in batches]

    m.set_option(_lib.DM_OPT_ASYNC, 1)
    nset = 3
    cap = max(len(b) for b in batches)
    host = [model.PinnedArray(cap * 28, gpu_device) for _ in range(nset)]
    dev = [model.DeviceArray((cap * 28,), np.uint8, gpu_device) for _ in range(nset)]
    cls = [model.DeviceArray((cap,), np.uint8, gpu_device) for _ in range(len(batches))]
    lib = _lib.load()
    for k, rows in enumerate(batches):
        i = k % nset
        m.wait_mark(i)                                   # never recorded for the first nset batches: returns at once
        host[i].view(np.float32, rows.size)[:] = rows.ravel()
        _lib.check(getattr(lib, copy)(m._h, dev[i].ptr, host[i].ptr, rows.nbytes))      # in stream order / on the copy stream, ahead of the queue
        m.predict_rows_device(dev[i].ptr, len(rows), 10, len(rows) - 20, cls[k].ptr + 10)
        m.mark(i)
    for i in range(nset):
        m.wait_mark(i)
    for k, rows in enumerate(batches):
        got = cls[k].to_host()[10:len(rows) - 10]
        assert np.array_equal(got, want[k]), k
    with pytest.raises(_lib.DeepModHipError):
        m.mark(8)
    with pytest.raises(_lib.DeepModHipError):
        m.wait_mark(-1)
    for blk in host + dev + cls:
        blk.free()
    m.close()


def test_range_violation_surfaces_at_the_marker(gpu_device):
    w = synth.synthetic_weights(22, 4.0)
    m = model.BiLSTMModel(w, device=gpu_device, precision="f16x3")
    m.set_option(_lib.DM_OPT_ASYNC, 1)
    rows = synth.synthetic_windows(400, seed=2)[:, 10, :].copy()
    rows[200, 0] = 1.0e6
    d = model.DeviceArray.from_host(rows, gpu_device)
    c = model.DeviceArray((len(rows),), np.uint8, gpu_device)
    m.predict_rows_device(d.ptr, len(rows), 10, len(rows) - 20, c.ptr + 10)
    m.mark(0)
    with pytest.raises(_lib.DeepModRangeError):
        m.wait_mark(0)
    m.wait_mark(0)                                       # reported once
    m.sync()
    m.close()


def test_reserved_cus_do_not_change_results(gpu_device):
    """DM_OPT_RESERVED_CUS shrinks the persistent grid of every classifier kernel; results are those of the full grid."""
    w = synth.synthetic_weights(26, 4.0)
    x = synth.synthetic_windows(70000, seed=9)             # more work items than workgroups for every kernel
    for prec in ("f16x3", "f32"):
        m = model.BiLSTMModel(w, device=gpu_device, precision=prec)
        p0, c0 = m.predict_windows(x)
        m.set_option(_lib.DM_OPT_RESERVED_CUS, 32)
        p1, c1 = m.predict_windows(x)
        assert np.array_equal(c0, c1) and np.array_equal(p0, p1), prec
        for bad in (-1, 100000):
            with pytest.raises(_lib.DeepModHipError):
                m.set_option(_lib.DM_OPT_RESERVED_CUS, bad)
        m.close()


def test_streaming_batches_outside_the_f16_range_take_the_fp32_kernel(tmp_path, gpu_device):
    """A read with an event length no f16 scheme can carry (1e9 samples) and a NaN feature: the streaming engine runs the
    batch that holds it with the fp32 kernel (host-side range check in the feeder) and its BED equals an all-fp32 run; the
    reference-shaped session adapter and the stored path's batched call fall back per call."""
    import os
    from deepmod_amd import detect, predstore, stream, synth_reads
    files = synth_reads.write_synthetic_run(str(tmp_path / 'in'), n_reads=8, reads_per_file=2, genome_len=6000, seed=3, chrom='chrA',
                                            min_len=200, max_len=500)
    pk = predstore.load_packed(files[1])
    tx = np.array(pk['tx'])
    tx[int(pk['row_off'][0]) + 150, 6] = 1.0e9
    tx[int(pk['row_off'][1]) + 160, 4] = np.nan
    reads = []
    for i, meta in enumerate(pk['reads']):
        ro, bo, eo = pk['row_off'], pk['bmi_off'], pk['ev_off']
        reads.append(dict(meta, tx=tx[ro[i]:ro[i + 1]], refbase=pk['refbase'][bo[i]:bo[i + 1]], readbase=pk['readbase'][bo[i]:bo[i + 1]],
                          refbasei=pk['refbasei'][bo[i]:bo[i + 1]], evbase=pk['evbase'][eo[i]:eo[i + 1]]))
    predstore.save_packed_container(files[1], reads, pk['contig_len'])
    prefix = str(tmp_path / 'model' / 'm')
    os.makedirs(os.path.dirname(prefix))
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    mo = {'fnum': 7, 'hidden': 100, 'windowsize': 21, 'modfile': [prefix, os.path.dirname(prefix) + '/'], 'outFolder': str(tmp_path / 'out'),
          'Base': 'C'}
    os.makedirs(mo['outFolder'])
    beds = {}
    for name, env in (('default', None), ('f32', 'f32')):
        if env:
            os.environ['DEEPMOD_PRECISION'] = env
        try:
            backend = stream.HipBackend(mo, gpu_device)
            eng = stream.StreamEngine(mo, backend)
            eng.run(iter([files[:2], files[2:]]), feeders=1)
            beds[name] = {k: bytes(v) for k, v in eng.finalize(None, None, write=False).items()}
            n_f32 = eng.stats['submit_f32_batches']
            backend.close()
        finally:
            os.environ.pop('DEEPMOD_PRECISION', None)
        assert n_f32 == (1 if name == 'default' else 0)          # only the batch with the poisoned container switches kernels
    assert beds['default'] == beds['f32'] and len(beds['f32']) > 0

    # the per-call fallbacks: dm_predict_read on the poisoned rows, and the sess.run seam on materialised windows
    from deepmod_amd import model as dm
    w = synth.synthetic_weights(26, 4.0)
    m = dm.BiLSTMModel(w, device=gpu_device)
    rows = np.ascontiguousarray(tx[:600], np.float32)
    with pytest.raises(_lib.DeepModRangeError):
        m.predict_read(rows, 10, len(rows) - 20)
    got = detect.predict_rows_any_range(m, rows, 10, len(rows) - 20)
    assert m.get_info(_lib.DM_INFO_PRECISION) == _lib.DM_PREC_F16X3          # restored
    m.set_precision('f32')
    assert np.array_equal(got, m.predict_read(rows, 10, len(rows) - 20, want_prob=False)[1])
    m.close()


def test_rows_assembled_on_the_device_equal_the_host_rows(tmp_path, gpu_device):
    """Round 5: a batch of raw reads is handed over in the device form - (mean, stdv, length) per event, a
    class byte per row, a descriptor per read - and dm_rows_assemble builds get_Feature's [R][7] matrix (myDetect.py:839-903) on the device:
    bit for bit the rows dm_rows_emit writes on the host, from 13 instead of 28 bytes per row; the streaming engine gives the same BED either way."""
    from deepmod_amd import _lib, model as dm, signal as dmsignal
    files, fasta = synth_reads.write_synthetic_raw_run(str(tmp_path / 'in'), n_reads=30, reads_per_file=6, genome_len=40000, seed=8, chrom='chrR',
                                                       min_len=300, max_len=2500)
    prefix = str(tmp_path / 'model' / 'm')
    os.makedirs(os.path.dirname(prefix))
    w = synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    mo = {'fnum': 7, 'hidden': 100, 'windowsize': 21, 'modfile': [prefix, os.path.dirname(prefix) + '/'], 'outFolder': str(tmp_path / 'out'), 'Base': 'C',
          'Ref': fasta, 'alignStr': 'minimap2', 'region': [[None, None, None]], 'ConUnk': True, 'SignalGroup': 'simple', 'outLevel': 3, 'device': gpu_device}
    os.makedirs(mo['outFolder'])
    norm = dmsignal.SignalNormalizer(gpu_device)
    dev = stream._prepare_batch_c(dict(mo), files, lambda: norm)
    host = stream._prepare_batch_c(dict(mo, rows_on_device=False), files, lambda: norm)
    assert dev.ev3 is not None and dev._rows is None and host.ev3 is None and dev.n_rows == host.n_rows > 20000
    R, E, NR = dev.n_rows, len(dev.ev3), len(dev.rdesc)
    m = dm.BiLSTMModel(w, device=gpu_device)
    d_ev3, d_code, d_rdesc = (dm.DeviceArray.from_host(a, gpu_device) for a in (dev.ev3, dev.code, dev.rdesc))
    d_rows = dm.DeviceArray((R, 7), np.float32, gpu_device)
    m.assemble_rows_device(d_rows.ptr, d_code.ptr, d_ev3.ptr, d_rdesc.ptr, NR, R)
    m.sync()
    got = d_rows.to_host()
    assert np.array_equal(got.view(np.uint32), host._rows.view(np.uint32))                # the device's rows == the host's rows, bit for bit
    assert np.array_equal(stream.assemble_rows(dev.ev3, dev.code, dev.rdesc, R).view(np.uint32), got.view(np.uint32))      # == the host restatement
    assert dev.ev3.nbytes + dev.code.nbytes + dev.rdesc.nbytes < 0.5 * host._rows.nbytes
    for a in (d_ev3, d_code, d_rdesc, d_rows):
        a.free()
    m.close()
    beds = {}
    for name, on_dev in (('device', True), ('host', False)):
        mo2 = dict(mo, rows_on_device=on_dev)
        backend = stream.HipBackend(mo2, gpu_device)
        eng = stream.StreamEngine(mo2, backend)
        eng.run(iter([files[:3], files[3:]]), feeders=1, make_normalizer=lambda: norm)
        beds[name] = {k: bytes(v) for k, v in eng.finalize(None, None, write=False).items()}
        assert (backend.timing['rows_on_device'] > 0) == on_dev
        backend.close()
    assert beds['device'] == beds['host'] and len(beds['device']) == 2 and all(len(v) > 1000 for v in beds['device'].values())
    norm.close()


@pytest.mark.gpu
def test_device_block_pool_reuses_the_smallest_block_that_fits(gpu_device):
    """stream.DeviceBlockPool: the pooled device blocks of resident signal requests - a block is allocated only when no free one is large enough, the smallest
    free block that fits is the one handed out, blocks given back are used again, close() frees what is free."""
    pool = stream.DeviceBlockPool(gpu_device)
    a = pool.take(1 << 20)
    b = pool.take(8 << 20)
    assert pool.allocated == 2 and a.nbytes >= (1 << 20) and b.nbytes >= (8 << 20) and a.ptr != b.ptr
    pool.give(b)
    pool.give(a)
    pool.give(None)
    c = pool.take(1 << 19)                 # the 1 MB block, not the 8 MB one
    assert c.ptr == a.ptr and pool.allocated == 2
    d = pool.take(4 << 20)
    assert d.ptr == b.ptr and pool.allocated == 2
    e = pool.take(4 << 20)                 # nothing free: a new one
    assert pool.allocated == 3 and e.ptr not in (a.ptr, b.ptr)
    for blk in (c, d, e):
        pool.give(blk)
    pool.close()
    f = pool.take(16)                      # a closed pool starts over
    assert f.nbytes >= 16 and pool.allocated == 4
    pool.give(f)
    pool.close()
