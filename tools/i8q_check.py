import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from deepmod_amd import _lib, model, synth
from oracle import oracle_np
from conftest import trained_like_weights
for name, w in (('scale1', synth.synthetic_weights(21, 1.0)), ('scale4', synth.synthetic_weights(26, 4.0)), ('trained', trained_like_weights())):
    ms = {}
    for sh in (16, 32):
        m = model.BiLSTMModel(w, 0, precision='f16i8'); m.set_option(_lib.DM_OPT_F16X3_SHAPE, sh); ms[sh] = m
    for n in (1, 17, 129, 1000, 20000):
        x = synth.synthetic_windows(n, seed=100 + n)
        ref, rc = oracle_np.predict_windows_c(w, x)
        line = "%-8s n %6d:" % (name, n)
        for sh, m in ms.items():
            p, c = m.predict_windows(x)
            near = np.abs(ref[:, 1] - 0.5) < 2e-4
            line += "  shape %d: max|dp| %.3g flips %d" % (sh, np.abs(p - ref).max(), int(((c.astype(np.int64) != rc) & ~near).sum()))
        print(line, flush=True)
w = synth.synthetic_weights(26, 4.0)
n = 65536
dx = model.DeviceArray.from_host(synth.synthetic_windows(n, seed=1), 0)
dc = model.DeviceArray((n,), np.uint8, 0)
for rep in range(2):
    for prec, sh in (('f16i8', 32), ('f16i8', 16), ('f16x3', 16)):
        m = model.BiLSTMModel(w, 0, precision=prec); m.set_option(_lib.DM_OPT_F16X3_SHAPE, sh)
        m.set_option(_lib.DM_OPT_PROFILE, 1); m.set_option(_lib.DM_OPT_ASYNC, 1)
        for _ in range(40): m.predict_windows(dx, cls=dc, want_prob=False)
        m.sync(); m.profile_reset()
        t0 = time.time()
        while time.time() - t0 < 2.0:
            for _ in range(50): m.predict_windows(dx, cls=dc, want_prob=False)
            m.sync()
        t, l, _ = m.profile_get()
        print("%s shape %d: %.4f ms per launch" % (prec, sh, t / l), flush=True)
        m.close()
