"""Golden vectors for raw-signal normalisation + per-event statistics, produced by running the REFERENCE's own
functions (/root/reference/bin/DeepMod_scripts/myDetect.py: getFast5Info :297-343, which calls mnormalized :266-282)
in the build container with stub `tensorflow` / `h5py` modules and an in-memory stand-in for the FAST5 reader.

Output (plain data): host_signal.npz — per case: raw int16 signal, event start/length, and the reference's results:
m_event mean/stdv (float32) after the loop, the number of events kept, the normalised signal.

Run only here (needs /root/reference):  python tests/golden/make_golden_signal.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_host import EVENT_DTYPE, import_reference  # noqa: E402


class _DS:
    def __init__(self, v):
        self.v = v

    def __getitem__(self, k):
        return self.v


class _Reader(dict):
    pass


def run_reference(myDetect, raw, start, length):
    ev = np.zeros(len(start), dtype=EVENT_DTYPE)
    ev['start'] = start
    ev['length'] = length
    ev['model_state'] = 'NNANN'
    myDetect.get_channel_info = lambda mo, sp: sp.__setitem__('channel_info', {'ok': 1})
    myDetect.getAlbacoreVersion = lambda mo, sp: sp.__setitem__('used_albacore_version', 2)
    myDetect.getRawInfo = lambda mo, sp: sp.__setitem__('raw_signals', raw.copy())
    myDetect.getEvent = lambda mo, sp: sp.__setitem__('m_event', ev)
    mo = {'basecall_1d': 'Basecall_1D_000', 'basecall_2strand': 'BaseCalled_template', 'outLevel': 2}
    fq_path = ''.join([myDetect.fast5_analysis, '/', mo['basecall_1d'], '/', mo['basecall_2strand'], '/', myDetect.fast5_basecall_fq])
    sp = {'mfile_path': 'synthetic.fast5', 'f5status': '', 'f5reader': _Reader({fq_path: _DS(b"@read1\nACGT\n+\n!!!!\n")})}
    myDetect.getFast5Info(mo, sp)
    assert sp['f5status'] == ''
    return sp


def make_case(rng, n_raw, first, mean_len, loc=480.0, scale=70.0, long_at=None, long_len=0, overrun=None, outliers=0):
    raw = np.clip(np.round(rng.normal(loc, scale, n_raw)), -32768, 32767).astype(np.int16)
    if outliers:
        idx = rng.integers(0, n_raw, outliers)
        raw[idx] = rng.choice(np.array([-30000, -9000, 25000, 32000, 1500, -20], np.int16), outliers)
    lens = []
    pos = first
    while True:
        ln = int(rng.geometric(1.0 / mean_len))
        if long_at is not None and len(lens) == long_at:
            ln = long_len
        if pos + ln > n_raw - 5:
            break
        lens.append(ln)
        pos += ln
    length = np.array(lens, np.uint64)
    start = (first + np.concatenate([[0], np.cumsum(length[:-1])])).astype(np.uint64)
    if overrun == 'clamp':          # last event runs past the end of the signal: numpy clamps the slice
        length[-1] = np.uint64(n_raw - int(start[-1]) + 40)
    elif overrun is not None:       # event `overrun` and everything after it start beyond the signal
        start[overrun:] += np.uint64(n_raw)
    return raw, start, length


def main():
    myDetect = import_reference()
    rng = np.random.default_rng(20260928)
    cases = {
        'typical': make_case(rng, 30011, 137, 9.0),
        'even_slice': make_case(rng, 20000, 0, 7.0, loc=100.0, scale=3.0),              # tiny value range: x.5 medians
        'long_events': make_case(rng, 60000, 50, 12.0, long_at=20, long_len=20011),      # > 8192 and > 128 samples
        'mid_events': make_case(rng, 16000, 3, 150.0),                                    # 8 <= n <= 128 and recursion
        'outliers': make_case(rng, 25000, 11, 9.0, outliers=400),                         # values outside the LDS bins
        'clamped_tail': make_case(rng, 12000, 20, 9.0, overrun='clamp'),
        'empty_late': make_case(rng, 12000, 20, 9.0, overrun=700),                        # first empty event i > 500
        'empty_early': make_case(rng, 12000, 20, 9.0, overrun=300),                       # i <= 500: table kept
    }
    out = {}
    for name, (raw, start, length) in cases.items():
        sp = run_reference(myDetect, raw, start, length)
        ev = sp['m_event']
        out[name + '.raw'] = raw
        out[name + '.start'] = start
        out[name + '.length'] = length
        out[name + '.n_kept'] = np.int64(len(ev))
        out[name + '.mean'] = ev['mean'].astype(np.float32)
        out[name + '.stdv'] = ev['stdv'].astype(np.float32)
        if name in ('typical', 'outliers'):      # the normalised signal itself (float64) for two cases
            out[name + '.signal'] = np.asarray(sp['raw_signals'], np.float64)
        print(name, len(raw), 'samples', len(start), 'events ->', len(ev), 'kept; mean[0..3]', ev['mean'][:3], 'stdv', ev['stdv'][:3])
    np.savez_compressed(os.path.join(HERE, 'host_signal.npz'), **out)


if __name__ == '__main__':
    main()
