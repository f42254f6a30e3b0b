"""Raw-signal normalisation and per-event statistics on the GPU.

Host-side mirror of the reference's `mnormalized` (myDetect.py:266-282) and of the per-event mean / stdv loop
at the end of `getFast5Info` (:332-343): same `sp_param` keys in, same fields of `sp_param['m_event']` updated.
The arithmetic runs in libdeepmod_hip.so (deepmod_amd/csrc/signal.hip.inc); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _lib


class SignalNormalizer:
    """One dm_signal handle (device buffers + stream) on one GPU; reuse it across reads."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        self._h = self._lib.dm_signal_create(device)
        if not self._h:
            raise _lib.DeepModHipError("dm_signal_create: " + _lib.last_error())

    def close(self):
        if self._h:
            self._lib.dm_signal_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def event_stats(self, raw, ev_start, ev_length, want_signal: bool = False):
        """raw int16[n]; ev_start/ev_length uint64[E] -> (mean f32[E], stdv f32[E], norm dict, first_empty, signal|None)"""
        raw = np.ascontiguousarray(raw)
        if raw.dtype != np.int16:
            if not np.issubdtype(raw.dtype, np.integer) or raw.size and (raw.min() < -32768 or raw.max() > 32767):
                raise ValueError("raw signal must hold int16 DAC values (FAST5 Raw/Reads/*/Signal)")
            raw = raw.astype(np.int16)
        st = np.ascontiguousarray(ev_start, dtype=np.uint64)
        ln = np.ascontiguousarray(ev_length, dtype=np.uint64)
        n_ev = len(st)
        mean = np.empty(n_ev, np.float32)
        stdv = np.empty(n_ev, np.float32)
        norm6 = np.empty(6, np.float64)
        first_empty = ctypes.c_int64(0)
        sig = np.empty(len(raw), np.float64) if want_signal else None
        _lib.check(self._lib.dm_signal_event_stats(
            self._h, raw.ctypes.data, len(raw), st.ctypes.data, ln.ctypes.data, n_ev, mean.ctypes.data, stdv.ctypes.data,
            norm6.ctypes.data, ctypes.byref(first_empty), sig.ctypes.data if want_signal else None))
        norm = dict(zip(("mshift", "mscale", "read_med", "read_mad", "lower_lim", "upper_lim"), norm6.tolist()))
        return mean, stdv, norm, int(first_empty.value), sig


    def event_stats_arrays(self, raw_parts, raw_off, ev_start, ev_length, ev_off):
        """The batched call on arrays that are already laid out back to back (stream._prepare_batch_c): raw_parts - int16 arrays
        whose concatenation is the samples of all reads, raw_off / ev_off [n + 1] int64, ev_start / ev_length uint64 (starts relative
        to the read's first sample).  -> (mean f32[E], stdv f32[E], first_empty int64[n])"""
        n = len(raw_off) - 1
        raw = raw_parts[0] if len(raw_parts) == 1 else np.concatenate(raw_parts)
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        st = np.ascontiguousarray(ev_start, dtype=np.uint64)
        ln = np.ascontiguousarray(ev_length, dtype=np.uint64)
        raw_off = np.ascontiguousarray(raw_off, np.int64)
        ev_off = np.ascontiguousarray(ev_off, np.int64)
        mean = np.empty(len(st), np.float32)
        stdv = np.empty(len(st), np.float32)
        norm6 = np.empty((n, 6), np.float64)
        first_empty = np.empty(n, np.int64)
        _lib.check(self._lib.dm_signal_event_stats_batch(self._h, n, raw.ctypes.data, raw_off.ctypes.data, st.ctypes.data, ln.ctypes.data,
                                                         ev_off.ctypes.data, mean.ctypes.data, stdv.ctypes.data, norm6.ctypes.data,
                                                         first_empty.ctypes.data))
        return mean, stdv, first_empty

    def event_stats_device(self, raw, raw_off, ev_start, ev_length, ev_off, block_ptr: int, fb_mean=None, fb_stdv=None):
        """The RESIDENT form (dm_signal_plan_batch + dm_signal_event_stats_device): the statistics of every merged event of the batch - (mean, stdv,
        length), the fall-back values fb_mean / fb_stdv merged in for events at or behind a read's first empty event - are written into the device
        block at block_ptr ([n_events][3] float32) and stay there.  -> (first_empty int64[n], range flag)"""
        n = len(raw_off) - 1
        raw = np.ascontiguousarray(raw, dtype=np.int16)
        st = np.ascontiguousarray(ev_start, dtype=np.uint64)
        ln = np.ascontiguousarray(ev_length, dtype=np.uint64)
        raw_off = np.ascontiguousarray(raw_off, np.int64)
        ev_off = np.ascontiguousarray(ev_off, np.int64)
        first_empty = np.empty(n, np.int64)
        _lib.check(self._lib.dm_signal_plan_batch(n, raw_off.ctypes.data, ev_off.ctypes.data, st.ctypes.data, ln.ctypes.data, first_empty.ctypes.data))
        fm = None if fb_mean is None else np.ascontiguousarray(fb_mean, np.float32)
        fs = None if fb_stdv is None else np.ascontiguousarray(fb_stdv, np.float32)
        flags = ctypes.c_int32(0)
        _lib.check(self._lib.dm_signal_event_stats_device(self._h, n, raw.ctypes.data, raw_off.ctypes.data, st.ctypes.data, ln.ctypes.data, ev_off.ctypes.data,
                                                          first_empty.ctypes.data, None if fm is None else fm.ctypes.data,
                                                          None if fs is None else fs.ctypes.data, block_ptr, None, ctypes.byref(flags)))
        return first_empty, int(flags.value)

    def event_stats_batch(self, reads):
        """reads: [(raw int16[n], ev_start uint64[E], ev_length uint64[E])] -> [(mean, stdv, norm dict, first_empty)] with one
        device round trip for the whole list (dm_signal_event_stats_batch; bit-identical to per-read event_stats)."""
        if not reads:
            return []
        raws = [np.ascontiguousarray(r[0], dtype=np.int16) for r in reads]
        sts = [np.ascontiguousarray(r[1], dtype=np.uint64) for r in reads]
        lns = [np.ascontiguousarray(r[2], dtype=np.uint64) for r in reads]
        raw_off = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int64)
        ev_off = np.concatenate([[0], np.cumsum([len(t) for t in sts])]).astype(np.int64)
        raw = np.concatenate(raws)
        st, ln = np.concatenate(sts), np.concatenate(lns)
        n = len(reads)
        mean = np.empty(len(st), np.float32)
        stdv = np.empty(len(st), np.float32)
        norm6 = np.empty((n, 6), np.float64)
        first_empty = np.empty(n, np.int64)
        _lib.check(self._lib.dm_signal_event_stats_batch(self._h, n, raw.ctypes.data, raw_off.ctypes.data, st.ctypes.data, ln.ctypes.data,
                                                         ev_off.ctypes.data, mean.ctypes.data, stdv.ctypes.data, norm6.ctypes.data,
                                                         first_empty.ctypes.data))
        keys = ("mshift", "mscale", "read_med", "read_mad", "lower_lim", "upper_lim")
        return [(mean[ev_off[i]:ev_off[i + 1]], stdv[ev_off[i]:ev_off[i + 1]], dict(zip(keys, norm6[i].tolist())), int(first_empty[i]))
                for i in range(n)]


_default: Optional[SignalNormalizer] = None


def mnormalized_event_stats(moptions, sp_param, normalizer: Optional[SignalNormalizer] = None, want_signal: bool = False):
    """`mnormalized(moptions, sp_param)` followed by the per-event statistics loop of `getFast5Info`
    (myDetect.py:266-282 + :332-343) in one device pass.

    In : sp_param['raw_signals'] int16[n], sp_param['m_event'] (start, length filled; mean, stdv overwritten).
    Out: sp_param['m_event']['mean'|'stdv'] (float32, rounded to 3 decimals), sp_param['norm'] (the four medians and
         the clip limits), sp_param['raw_signals'] replaced by the normalised float64 signal when want_signal.
    The reference's handling of an event whose slice is empty is kept: the loop stops there; if that is event i > 500
    the table is truncated to m_event[:i-1], otherwise it is left as is (myDetect.py:337-340; the status assignment on
    :340 is a comparison and has no effect)."""
    global _default
    if normalizer is None:
        if _default is None:
            _default = SignalNormalizer(int(moptions.get('device', 0)) if hasattr(moptions, 'get') else 0)
        normalizer = _default
    ev = sp_param['m_event']
    _check_event_span(sp_param)
    mean, stdv, norm, first_empty, sig = normalizer.event_stats(sp_param['raw_signals'], ev['start'], ev['length'], want_signal)
    return _apply_event_stats(sp_param, mean, stdv, norm, first_empty, sig if want_signal else None)


def _check_event_span(sp_param):
    ev = sp_param['m_event']
    if not ev['start'][0] < (ev['start'][-1] + ev['length'][-1]):
        print('Fatal error signal start position is less than the end position', sp_param.get('mfile_path'),
              ev['start'][0], ev['start'][-1], ev['length'][-1])


def _apply_event_stats(sp_param, mean, stdv, norm, first_empty, sig=None):
    """Write the device results into sp_param['m_event'] with the reference's handling of an empty event slice."""
    ev = sp_param['m_event']
    want_signal = sig is not None
    n_ok = first_empty
    ev['mean'][:n_ok] = mean[:n_ok]
    ev['stdv'][:n_ok] = stdv[:n_ok]
    if first_empty < len(ev):
        i = first_empty
        print('Signal out of range {}: {}-{} {};{} for {}'.format(i, ev['start'][i], ev['length'][i], len(ev),
                                                                  len(sp_param['raw_signals']), sp_param.get('mfile_path')))
        if i > 500:
            sp_param['m_event'] = ev[:i - 1]
    sp_param['norm'] = norm
    if want_signal:
        sp_param['raw_signals'] = sig
    return sp_param


def mnormalized_event_stats_batch(moptions, sp_params, normalizer: Optional[SignalNormalizer] = None):
    """mnormalized_event_stats for a list of reads with one device round trip (reads of a worker batch travel together: a
    single 120 k-sample read is launch / latency bound).  Same sp_param contract per read; a read the batched call
    cannot take (events covering no signal) makes the whole list fall back to per-read calls, so that only that read fails."""
    global _default
    if normalizer is None:
        if _default is None:
            _default = SignalNormalizer(int(moptions.get('device', 0)) if hasattr(moptions, 'get') else 0)
        normalizer = _default
    for sp in sp_params:
        _check_event_span(sp)
    try:
        res = normalizer.event_stats_batch([(sp['raw_signals'], sp['m_event']['start'], sp['m_event']['length']) for sp in sp_params])
    except _lib.DeepModHipError:
        res = None
    out = []
    for i, sp in enumerate(sp_params):
        try:
            if res is None:
                mnormalized_event_stats(moptions, sp, normalizer)
            else:
                _apply_event_stats(sp, *res[i])
            out.append(None)
        except Exception as exc:          # reported per read by the caller
            out.append(exc)
    return out
