"""ORACLE — TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).  Never imported by deepmod_amd/.

numpy restatement of the reference's raw-signal normalisation and per-event statistics:
  /root/reference/bin/DeepMod_scripts/myDetect.py:266-282  mnormalized()
  /root/reference/bin/DeepMod_scripts/myDetect.py:332-343  per-event round(np.mean, 3) / round(np.std, 3)
The reference code IS numpy, so the restatement calls the same numpy functions on the same slices; the only
liberty is a vectorised clip instead of the per-sample Python conditional (:282), which selects the same values.

PARITY PIN: tests/golden/host_signal.npz holds inputs and outputs of the reference's own functions executed here
under stub modules (tests/golden/make_golden_signal.py); tests/test_oracle_golden.py checks this file against it.
"""
from __future__ import annotations

import numpy as np


def mnormalized(raw_signals, m_event):
    """-> (normalised float64 signal, dict of the four medians and the limits)   [myDetect.py:266-282]"""
    lo = int(m_event['start'][0])
    hi = int(m_event['start'][-1] + m_event['length'][-1])
    raw = np.asarray(raw_signals)
    mshift = np.median(raw[lo:hi])
    mscale = np.median(np.abs(raw[lo:hi] - mshift))
    sig = (raw - mshift) / mscale
    read_med = np.median(sig[lo:hi])
    read_mad = np.median(np.abs(sig[lo:hi] - read_med))
    lower_lim = read_med - (read_mad * 5)
    upper_lim = read_med + (read_mad * 5)
    sig = np.round(np.where(sig > upper_lim, upper_lim, np.where(sig < lower_lim, lower_lim, sig)), 3)
    return sig, dict(mshift=float(mshift), mscale=float(mscale), read_med=float(read_med), read_mad=float(read_mad),
                     lower_lim=float(lower_lim), upper_lim=float(upper_lim))


def event_stats(sig, m_event):
    """-> (mean f32[E], stdv f32[E], first_empty)   [myDetect.py:332-343]"""
    n = len(m_event)
    mean = np.full(n, np.nan, np.float32)
    stdv = np.full(n, np.nan, np.float32)
    first_empty = n
    for i in range(n):
        seg = sig[int(m_event['start'][i]):int(m_event['start'][i] + m_event['length'][i])]
        if len(seg) == 0:
            first_empty = i
            break
        mean[i] = round(np.mean(seg), 3)
        stdv[i] = round(np.std(seg), 3)
    return mean, stdv, first_empty
