"""TEST INFRASTRUCTURE (run by tests/test_gpu_stress.py in a subprocess): the stream orderings of the library under stress.

The library juggles a model stream, per-summary streams, dm_summary_follow, the communicator's stream, signal handles with their
own streams and stream markers.  One iteration = create a summary (every few iterations a 3e8-position one: 3.6 GB of counters
whose clearing takes a millisecond - the race of round 2 was a clear that had not landed when the first kernel ran), follow the
model, queue classify + accumulate WITHOUT waiting, grow the counters, reduce on a 1-rank RCCL communicator, reduce-scatter, fetch,
compare with the oracle, destroy - while two threads keep two signal handles busy on their own streams.
Prints one JSON line: iterations, a digest of every fetched result (same bits under AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING).

    python tests/stress_stream_order.py <iterations> <big_every> <rendezvous dir>
"""
import hashlib
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmod_amd import _lib, comm, model, signal as dmsignal, summary, synth      # noqa: E402
from oracle import oracle_np                                                          # noqa: E402


def main():
    iters, big_every, rdv_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dev = 0
    w = synth.synthetic_weights(26, 4.0)
    n = 20000
    x = synth.synthetic_windows(n, seed=31)
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
    clear = np.abs(ref_prob[:, 1] - 0.5) > 1e-4
    m = model.BiLSTMModel(w, device=dev)
    m.set_option(_lib.DM_OPT_ASYNC, 1)
    dx = model.DeviceArray.from_host(x, dev)
    dcls = model.DeviceArray((n,), np.uint8, dev)
    c = comm.Communicator.from_rendezvous(dev, comm.FileRendezvous(rdv_dir, 0, 1))
    rng = np.random.default_rng(5)

    # two signal handles busy on their own streams for the whole run
    stop = threading.Event()
    sig_err = []

    def signal_loop(seed):
        try:
            r = np.random.default_rng(seed)
            norm = dmsignal.SignalNormalizer(dev)
            reads = []
            for _ in range(6):
                raw = r.integers(300, 900, 60000).astype(np.int16)
                length = r.integers(3, 12, 6000).astype(np.uint64)
                start = np.concatenate([[5], 5 + np.cumsum(length)[:-1]]).astype(np.uint64)
                reads.append((raw, start, length))
            first = norm.event_stats_batch(reads)
            while not stop.is_set():
                again = norm.event_stats_batch(reads)
                for a, b in zip(first, again):
                    if not (np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True)):
                        raise AssertionError('signal statistics changed between calls')
            norm.close()
        except BaseException as exc:
            sig_err.append(repr(exc))

    threads = [threading.Thread(target=signal_loop, args=(s,), daemon=True) for s in (1, 2)]
    for t in threads:
        t.start()

    digest = hashlib.sha256()
    for it in range(iters):
        big = big_every > 0 and it % big_every == 0
        length = 300_000_000 if big else int(rng.integers(2_000_000, 4_000_000))
        grow_to = length + int(rng.integers(1, 1_000_000))
        base = int(rng.integers(0, length - n))
        pos = (base + np.sort(rng.integers(0, n // 4, n))).astype(np.int64)          # runs of equal positions: the ballot path
        flags = rng.integers(0, 4, n).astype(np.uint8)
        dpos = model.DeviceArray.from_host(pos, dev)
        dflags = model.DeviceArray.from_host(flags, dev)
        s = summary.PositionSummary(length, dev)
        s.follow(m)
        m.predict_windows(dx, cls=dcls, want_prob=False)              # queued, not waited for
        s.add_classified(dpos, dflags, dcls, n)                        # queued behind it on the same stream
        far = np.array([grow_to - 1], np.int64)
        s.grow(grow_to)                                                # waits for both, reallocates, copies
        s.add(far, np.array([3], np.uint8))
        s.reduce(c, 0)
        assert s.reduce_scatter(c) == (0, grow_to)
        if big:
            touch, cov, mod = s.fetch_slice()
        else:
            touch, cov, mod = s.fetch()
        # oracle counters at the touched positions
        f = (flags & 3) | (ref_cls.astype(np.uint8) << 2)
        up, inv = np.unique(pos, return_inverse=True)
        exp = [np.bincount(inv, weights=((f & msk) == msk), minlength=len(up)).astype(np.int64) for msk in (1, 3, 7)]
        ok_rows = np.ones(len(up), bool)
        bad_rows = np.unique(inv[~clear & ((f & 3) == 3)])                # positions that a near-tie window contributes to: mod may differ
        ok_rows[bad_rows] = False
        assert np.array_equal(touch[up], exp[0]) and np.array_equal(cov[up], exp[1]), 'iteration %d: touch / cov differ' % it
        assert np.array_equal(mod[up][ok_rows], exp[2][ok_rows]), 'iteration %d: mod differs' % it
        assert touch[grow_to - 1] == 1 and cov[grow_to - 1] == 1 and mod[grow_to - 1] == 0
        assert int(touch.sum(dtype=np.int64)) == int(exp[0].sum()) + 1 and int(cov.sum(dtype=np.int64)) == int(exp[1].sum()) + 1, \
            'iteration %d: counts outside the touched positions' % it
        for a in (touch[up], cov[up], mod[up]):
            digest.update(np.ascontiguousarray(a).tobytes())
        s.close()
        dpos.free()
        dflags.free()
    stop.set()
    for t in threads:
        t.join(timeout=60)
    assert not sig_err, sig_err
    c.close()
    m.close()
    print(json.dumps({"iterations": iters, "digest": digest.hexdigest()}), flush=True)


if __name__ == "__main__":
    main()
