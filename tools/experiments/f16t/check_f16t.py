"""Dev tool (GPU box): parity of the experimental tile-major kernel against the oracle + its speed next to the product kernel,
on a dev library built by trace_f16t.py.    DM_TAG=<tag> python tools/experiments/f16t/check_f16t.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
from deepmod_amd import _lib, model, synth
from oracle import oracle_np
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_abl", "lib_f16t_trace%s.so" % os.environ.get("DM_TAG", ""))
for scale in (1.0, 4.0):
    w = synth.synthetic_weights(21, scale)
    m = model.BiLSTMModel(w, 0, precision="f16x3t")
    x = synth.synthetic_windows(3000, seed=5)
    x[:50, :, 6] = 3000.0
    x[50:60, 3, 6] = 1.0e6
    ref_prob, ref_cls = oracle_np.predict_windows_c(w, x)
    p, c = m.predict_windows(x)
    near = np.abs(ref_prob[:, 1] - 0.5) < 1e-4
    print("scale %.0f: f16x3t max|dp| %.3g flips %d" % (scale, np.abs(p - ref_prob).max(), int(((c != ref_cls) & ~near).sum())), flush=True)
    m.close()
w = synth.synthetic_weights(26, 4.0)
n = 65536
x = synth.synthetic_windows(n, seed=1)
for name in ("f16x3", "f16x3t"):
    m = model.BiLSTMModel(w, 0, precision=name)
    m.set_option(_lib.DM_OPT_PROFILE, 1)
    dx = model.DeviceArray.from_host(x, 0); dc = model.DeviceArray((n,), np.uint8, 0)
    for _ in range(10): m.predict_windows(dx, cls=dc, want_prob=False)
    m.profile_reset()
    for _ in range(40): m.predict_windows(dx, cls=dc, want_prob=False)
    ms, launches, _ = m.profile_get()
    print("%-7s %.3f ms per 65,536 windows" % (name, ms / launches), flush=True)
    m.close()
