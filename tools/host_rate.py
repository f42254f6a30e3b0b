"""Dev tool: PCIe-inclusive rates of the host-buffer entry points (run on the GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import model, synth
w = synth.synthetic_weights(7, 1.0)
m = model.BiLSTMModel(w, 0)
n = 16 * 65536
x = synth.synthetic_windows(n, seed=1)
m.predict_windows(x[:65536])
t = time.perf_counter(); prob, cls = m.predict_windows(x); dt = time.perf_counter() - t
print("host windows  (588 B/window over PCIe): %.3g windows/s" % (n / dt))
dx = model.DeviceArray.from_host(x, 0); dc = model.DeviceArray((n,), np.uint8, 0)
t = time.perf_counter(); m.predict_windows(dx, cls=dc, want_prob=False); dt = time.perf_counter() - t
print("device-resident windows:                 %.3g windows/s" % (n / dt))
rows = np.zeros((n + 200, 7), np.float32); rows[100:-100] = x[:, 10, :]
m.predict_read(rows, 100, 65536)
t = time.perf_counter(); m.predict_read(rows, 100, n, want_prob=False); dt = time.perf_counter() - t
print("host rows, on-device windowing (28 B/row): %.3g windows/s" % (n / dt))
for nr in (1000, 8000, 32768):
    t = time.perf_counter()
    for _ in range(10): m.predict_read(rows[:nr + 200], 100, nr, want_prob=False)
    dt = (time.perf_counter() - t) / 10
    print("one read of %5d bases per call:          %.3g windows/s  (%.2f ms/call)" % (nr, nr / dt, dt * 1e3))
