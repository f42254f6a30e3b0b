"""Dev tool (GPU box): keep one classifier kernel busy for a few seconds (for tools/power_trace.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import _lib, model, synth
if os.environ.get('DM_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['DM_LIB'])
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
w = synth.synthetic_weights(26, 4.0)
n = 65536
m = model.BiLSTMModel(w, 0, precision=prec)
m.set_option(_lib.DM_OPT_PROFILE, 1)
m.set_option(_lib.DM_OPT_ASYNC, 1)
dx = model.DeviceArray.from_host(synth.synthetic_windows(n, seed=1), 0)
dc = model.DeviceArray((n,), np.uint8, 0)
t0 = time.time()
while time.time() - t0 < secs:
    for _ in range(50):
        m.predict_windows(dx, cls=dc, want_prob=False)
    m.sync()
ms, launches, _ = m.profile_get()
print("%s: %d launches, %.3f ms per 65,536-window launch" % (prec, launches, ms / launches))
