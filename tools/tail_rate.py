import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from deepmod_amd import _lib, model, synth
w = synth.synthetic_weights(7, 1.0)
m = model.BiLSTMModel(w, 0); m.set_option(_lib.DM_OPT_PROFILE, 1)
for n in (38400, 65536, 70000, 100000, 131072, 140000):
    x = synth.synthetic_windows(n, seed=1); dx = model.DeviceArray.from_host(x, 0); dc = model.DeviceArray((n,), np.uint8, 0)
    for _ in range(20): m.predict_windows(dx, cls=dc, want_prob=False)
    m.profile_reset()
    for _ in range(20): m.predict_windows(dx, cls=dc, want_prob=False)
    ms, l, _ = m.profile_get()
    print("n=%7d (%4d tiles): %.3f ms/launch -> %.3g windows/s" % (n, (n + 127) // 128, ms / l, n / (ms / l) * 1e3), flush=True)
    dx.free(); dc.free()
