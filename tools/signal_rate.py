"""Dev tool (GPU box): throughput of the raw-signal stage (dm_signal_event_stats) for a typical read and a large one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import _lib, model, signal

nz = signal.SignalNormalizer(0)
lib = _lib.load()
for n_raw, mean_len in ((120_000, 9.0), (16_000_000, 9.0)):
    rng = np.random.default_rng(1)
    raw = np.clip(np.round(rng.normal(480, 70, n_raw)), -32768, 32767).astype(np.int16)
    lens = rng.geometric(1.0 / mean_len, int(n_raw / mean_len)).astype(np.uint64)
    start = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.uint64)
    keep = (start + lens) <= n_raw
    start, lens = start[keep], lens[keep]
    nz.event_stats(raw, start, lens)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        nz.event_stats(raw, start, lens)
    dt_host = (time.perf_counter() - t0) / reps
    d_raw = model.DeviceArray.from_host(raw, 0)
    mean = np.empty(len(start), np.float32); stdv = np.empty(len(start), np.float32)
    call = lambda: _lib.check(lib.dm_signal_event_stats(nz._h, d_raw.ptr, n_raw, start.ctypes.data, lens.ctypes.data, len(start),
                                                        mean.ctypes.data, stdv.ctypes.data, None, None, None))
    call()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dt_dev = (time.perf_counter() - t0) / reps
    print("%9d samples %8d events: host signal %.3f ms (%.2e samples/s) | device-resident signal %.3f ms (%.2e samples/s)" %
          (n_raw, len(start), dt_host * 1e3, n_raw / dt_host, dt_dev * 1e3, n_raw / dt_dev), flush=True)
    d_raw.free()
