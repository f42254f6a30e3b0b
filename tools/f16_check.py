"""Dev tool: parity + speed of the split-f16 kernels vs the oracle and the fp32 kernel (run on the GPU box).
DM_PRECS=f32,f16x3,f16x3r picks the kernels, DM_LIB a dev build of the library, DM_REPS the launches timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import _lib, model, synth
from oracle import oracle_np
if os.environ.get('DM_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['DM_LIB'])
precs = os.environ.get('DM_PRECS', 'f32,f16x3,f16x3r').split(',')
for scale in (1.0, 4.0):
    w = synth.synthetic_weights(21, scale)
    m = model.BiLSTMModel(w, 0)
    for n in (3000, 77):
        x = synth.synthetic_windows(n, seed=5)
        x[:50, :, 6] = 3000.0
        ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
        near = np.abs(ref_prob[:, 1] - 0.5) < 1e-4
        out = []
        for name in precs:
            m.set_precision(name)
            p, c = m.predict_windows(x)
            out.append("%s max|dp| %.3g flips %d finite %s" % (name, np.abs(p - ref_prob).max(), int(((c != ref_cls) & ~near).sum()), np.isfinite(p).all()))
        print("scale %.0f n %d: %s (near ties %d)" % (scale, n, " | ".join(out), int(near.sum())), flush=True)
    m.close()
w = synth.synthetic_weights(26, 4.0)
n = 65536
x = synth.synthetic_windows(n, seed=1)
for rnd in range(int(os.environ.get('DM_ROUNDS', '2'))):
    for name in precs:
        m = model.BiLSTMModel(w, 0)
        m.set_precision(name)
        m.set_option(_lib.DM_OPT_PROFILE, 1)
        dx = model.DeviceArray.from_host(x, 0); dc = model.DeviceArray((n,), np.uint8, 0)
        m.predict_windows(dx, cls=dc, want_prob=False)
        m.profile_reset()
        for _ in range(int(os.environ.get('DM_REPS', '30'))): m.predict_windows(dx, cls=dc, want_prob=False)
        ms, launches, _ = m.profile_get()
        print("%-6s %.3f ms per 65,536 windows -> %.3g windows/s" % (name, ms / launches, n / (ms / launches) * 1e3), flush=True)
        m.close()
