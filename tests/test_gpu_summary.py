"""GPU (-m gpu): dense per-position summary (integer, bit-exact) vs the oracle restatement of
sum_handler's accumulation (myDetect.py:1089-1100), plus the 1-rank RCCL reduce."""
import numpy as np
import pytest

from deepmod_amd import model, summary
from oracle import oracle_np

pytestmark = pytest.mark.gpu


def _random_bases(n, length, seed):
    rng = np.random.default_rng(seed)
    pos = rng.integers(0, length, n).astype(np.int64)
    pos[: n // 2] = np.sort(pos[: n // 2])            # read-like sorted runs
    pos[n // 2: n // 2 + 5000] = 12345 % length        # heavy collisions on one position
    flags = rng.integers(0, 8, n).astype(np.uint8)
    return pos, flags


def _oracle(length, pos, flags):
    t = np.zeros(length, np.int32); c = np.zeros(length, np.int32); m = np.zeros(length, np.int32)
    oracle_np.summary_add_c(t, c, m, pos, flags)
    return t, c, m


@pytest.mark.parametrize("n,length", [(1, 10), (1000, 97), (200000, 50000), (3000000, 4641652)])
def test_summary_matches_oracle_bit_exact(n, length, gpu_device):
    pos, flags = _random_bases(n, length, seed=n)
    s = summary.PositionSummary(length, gpu_device)
    half = n // 2
    s.add(pos[:half], flags[:half])      # two calls: accumulation persists across calls
    s.add(pos[half:], flags[half:])
    got = s.fetch()
    want = _oracle(length, pos, flags)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    s.close()


def test_summary_classified_device_buffers(gpu_device):
    n, length = 65536, 100000
    pos, flags = _random_bases(n, length, seed=3)
    cls = (np.random.default_rng(9).random(n) < 0.3).astype(np.uint8)
    s = summary.PositionSummary(length, gpu_device)
    s.add_classified(model.DeviceArray.from_host(pos, gpu_device), model.DeviceArray.from_host(flags, gpu_device),
                     model.DeviceArray.from_host(cls, gpu_device), n)
    got = s.fetch()
    want = _oracle(length, pos, (flags & 3) | (cls << 2))
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_summary_rejects_out_of_range(gpu_device):
    from deepmod_amd import _lib
    s = summary.PositionSummary(100, gpu_device)
    with pytest.raises(_lib.DeepModHipError):
        s.add(np.array([5, 100], np.int64), np.array([1, 1], np.uint8))
    t, c, m = s.fetch()
    assert t[5] == 1 and t.sum() == 1


def test_empty_add_is_noop(gpu_device):
    s = summary.PositionSummary(10, gpu_device)
    s.add(np.zeros(0, np.int64), np.zeros(0, np.uint8))
    assert s.fetch()[0].sum() == 0


def test_persistent_communicator_single_rank(gpu_device, tmp_path):
    """nranks = 1 exercises dlopen(librccl), the file rendezvous of the unique id, ONE communicator serving several
    reduces (ncclReduce to root 0 and ncclAllReduce), the barrier / max helpers and the statistics."""
    from deepmod_amd import comm
    rdv = comm.FileRendezvous(str(tmp_path / 'rdv'), 0, 1)
    c = comm.Communicator.from_rendezvous(gpu_device, rdv)
    sums = []
    for i, length in enumerate((1000, 70000, 1000)):
        s = summary.PositionSummary(length, gpu_device)
        pos, flags = _random_bases(5000, length, seed=i)
        s.add(pos, flags)
        before = s.fetch()
        s.reduce(c, 0 if i < 2 else -1)
        for b, a in zip(before, s.fetch()):
            assert np.array_equal(a, b)
        sums.append(s)
    c.barrier()
    assert c.max(3.5) == 3.5
    st = c.stats()
    assert st["collectives"] == 3 and st["bytes"] == 4 * 3 * (1000 + 70000 + 1000) and st["rccl_nranks"] == 1
    c.close()


def test_reduce_scatter_single_rank_and_sliced_bed(gpu_device, tmp_path):
    """dm_summary_reduce_scatter with nranks = 1 (ncclReduceScatter really runs; the slice is the whole contig) and the slice
    formatter: the text of [first, first + count) pieces concatenated equals the text of the whole contig."""
    from deepmod_amd import comm
    rdv = comm.FileRendezvous(str(tmp_path / 'rdv'), 0, 1)
    c = comm.Communicator.from_rendezvous(gpu_device, rdv)
    for i, length in enumerate((1001, 70003)):
        s = summary.PositionSummary(length, gpu_device)
        pos, flags = _random_bases(9000, length, seed=10 + i)
        s.add(pos, flags)
        whole = s.fetch()
        assert s.reduce_scatter(c) == (0, length)
        for a, b in zip(s.fetch_slice(), whole):
            assert np.array_equal(a, b)
        bed = summary.bed_lines('chrQ', '-', 'C', *whole)
        cuts = [0, length // 3, length // 3 + 1, length - 5, length]
        parts = [summary.bed_lines('chrQ', '-', 'C', *[a[lo:hi] for a in whole], first_pos=lo) for lo, hi in zip(cuts[:-1], cuts[1:])]
        assert b''.join(parts) == bed and len(bed) > 0
        s.close()
    st = c.stats()
    assert st["collectives"] == 2 and st["bytes"] == 12 * (1001 + 70003)
    c.close()


def test_streaming_finalize_through_the_scatter_merge_on_one_rank(gpu_device, tmp_path):
    """StreamEngine.finalize's multi-rank form (dm_summary_reduce_scatter on a real communicator, dm_summary_fetch_slice, the slice
    formatter, part files joined by rank 0) forced onto one rank: the BED files equal those of the single-rank form."""
    import os
    from deepmod_amd import comm, stream, synth, synth_reads
    files = synth_reads.write_synthetic_run(str(tmp_path / 'in'), n_reads=12, reads_per_file=3, genome_len=8000, seed=3, chrom='chrA',
                                            min_len=200, max_len=600)
    prefix = str(tmp_path / 'model' / 'm')
    os.makedirs(os.path.dirname(prefix))
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    beds = {}
    for tag in ('plain', 'scatter'):
        out = str(tmp_path / ('out_' + tag))
        os.makedirs(out)
        mo = {'fnum': 7, 'hidden': 100, 'windowsize': 21, 'modfile': [prefix, os.path.dirname(prefix) + '/'], 'outFolder': out, 'Base': 'C',
              'force_scatter_merge': tag == 'scatter'}
        backend = stream.HipBackend(mo, gpu_device)
        eng = stream.StreamEngine(mo, backend)
        eng.run(iter([files[:2], files[2:]]), feeders=1)
        if tag == 'plain':
            eng.finalize(None, None)
        else:
            c = comm.Communicator.from_rendezvous(gpu_device, comm.FileRendezvous(str(tmp_path / 'rdv'), 0, 1))
            eng.finalize(None, lambda s: s.reduce_scatter(c))
            assert c.stats()["collectives"] == 2                  # one per contig x strand
            c.close()
        backend.close()
        beds[tag] = {f: open(os.path.join(out, f), 'rb').read() for f in sorted(os.listdir(out))}
    assert sorted(beds['plain']) == ['mod_pos.chrA+.C.bed', 'mod_pos.chrA-.C.bed'] and beds['scatter'] == beds['plain']
    assert all(len(b) > 500 for b in beds['plain'].values())


def test_summary_grow_keeps_counts(gpu_device):
    s = summary.PositionSummary(1000, gpu_device)
    pos, flags = _random_bases(20000, 1000, seed=4)
    s.add(pos, flags)
    before = s.fetch()
    s.grow(5000)
    s.add(np.array([4999], np.int64), np.array([7], np.uint8))
    t, c, m = s.fetch()
    assert len(t) == 5000 and t[4999] == 1 and c[4999] == 1 and m[4999] == 1
    for b, a in zip(before, (t, c, m)):
        assert np.array_equal(a[:1000], b) and a[1000:4999].sum() == 0
    s.grow(10)        # never shrinks
    assert s.length == 5000


def test_bed_bytes_from_gpu_counts(gpu_device):
    s = summary.PositionSummary(50, gpu_device)
    #            pos flags: C covered+mod, C covered, C deletion only, non-C
    s.add(np.array([7, 7, 7, 9, 20], np.int64), np.array([7, 3, 3, 1, 6], np.uint8))
    t, c, m = s.fetch()
    bed = summary.bed_lines("chrS", "+", "C", t, c, m)
    assert bed == (b"chrS 7 8 C 3 + 7 8 0,0,0 3 33 1 \n"
                   b"chrS 9 10 C 0 + 9 10 0,0,0 0 0 0 \n")


def test_async_classify_accumulate_queue_equals_synchronous_calls(gpu_device):
    """DM_OPT_ASYNC + dm_summary_follow: classify -> accumulate -> classify ... on one in-order stream without host
    waits gives the same counters as the synchronous calls; a deferred out-of-range error surfaces at sync."""
    from deepmod_amd import _lib, model, summary, synth
    w = synth.synthetic_weights(26, 4.0)
    n, length = 20000, 50000
    rng = np.random.default_rng(3)
    xs = [synth.synthetic_windows(n, seed=50 + i) for i in range(3)]
    poss = [rng.integers(0, length, n).astype(np.int64) for _ in range(3)]
    flags = [(rng.integers(0, 4, n)).astype(np.uint8) for _ in range(3)]

    def run(async_mode):
        m = model.BiLSTMModel(w, 0)
        s = summary.PositionSummary(length, 0)
        dx = [model.DeviceArray.from_host(x, 0) for x in xs]
        dp = [model.DeviceArray.from_host(p, 0) for p in poss]
        df = [model.DeviceArray.from_host(f, 0) for f in flags]
        dc = model.DeviceArray((n,), np.uint8, 0)
        if async_mode:
            m.set_option(_lib.DM_OPT_ASYNC, 1)
            s.follow(m)
        for i in range(3):
            m.predict_windows(dx[i], cls=dc, want_prob=False)
            s.add_classified(dp[i], df[i], dc, n)
        m.sync()
        out = s.fetch()
        if async_mode:      # an out-of-range position is reported at the next sync, not at the add
            bad = model.DeviceArray.from_host(np.array([length + 5], np.int64), 0)
            one = model.DeviceArray.from_host(np.array([3], np.uint8), 0)
            s.add_classified(bad, one, one, 1)
            with pytest.raises(_lib.DeepModHipError):
                s.sync()
            s.follow(None)
        s.close()
        m.close()
        return out

    a, b = run(False), run(True)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    assert int(a[1].sum()) > 0
