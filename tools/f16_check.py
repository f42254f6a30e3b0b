"""Dev tool: parity + speed of the split-f16 path vs the oracle and the fp32 path (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import _lib, model, synth
from oracle import oracle_np
if os.environ.get('DM_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['DM_LIB'])
for scale in (1.0, 4.0):
    w = synth.synthetic_weights(21, scale)
    m = model.BiLSTMModel(w, 0)
    x = synth.synthetic_windows(3000, seed=5)
    x[:50, :, 6] = 3000.0
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
    m.set_option(_lib.DM_OPT_PRECISION, _lib.DM_PREC_F32)
    p32, c32 = m.predict_windows(x)
    m.set_option(_lib.DM_OPT_PRECISION, _lib.DM_PREC_F16X3)
    p16, c16 = m.predict_windows(x)
    m.set_option(_lib.DM_OPT_PRECISION, _lib.DM_PREC_F16X3T)
    p16t, c16t = m.predict_windows(x)
    near = np.abs(ref_prob[:, 1] - 0.5) < 1e-4
    print("scale %.0f: f32 max|dp| %.3g flips %d | f16x3 max|dp| %.3g flips %d | f16x3t max|dp| %.3g flips %d (near ties %d) finite %s" % (
        scale, np.abs(p32 - ref_prob).max(), int(((c32 != ref_cls) & ~near).sum()),
        np.abs(p16 - ref_prob).max(), int(((c16 != ref_cls) & ~near).sum()),
        np.abs(p16t - ref_prob).max(), int(((c16t != ref_cls) & ~near).sum()), int(near.sum()), np.isfinite(p16t).all()), flush=True)
    m.close()
w = synth.synthetic_weights(26, 4.0)
n = 65536
x = synth.synthetic_windows(n, seed=1)
for prec, name in ((_lib.DM_PREC_F32, "f32"), (_lib.DM_PREC_F16X3, "f16x3"), (_lib.DM_PREC_F16X3T, "f16x3t")):
    m = model.BiLSTMModel(w, 0)
    m.set_option(_lib.DM_OPT_PRECISION, prec)
    m.set_option(_lib.DM_OPT_PROFILE, 1)
    dx = model.DeviceArray.from_host(x, 0); dc = model.DeviceArray((n,), np.uint8, 0)
    m.predict_windows(dx, cls=dc, want_prob=False)
    m.profile_reset()
    for _ in range(int(os.environ.get('DM_REPS', '30'))): m.predict_windows(dx, cls=dc, want_prob=False)
    ms, launches, _ = m.profile_get()
    print("%-6s %.3f ms per 65,536 windows -> %.3g windows/s" % (name, ms / launches, n / (ms / launches) * 1e3))
    m.close()
