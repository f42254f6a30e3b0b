"""Dev tool (GPU box): max |dp| of the classifier kernels against the fp32 oracle and against the float64 value of the graph,
over weight seeds and scales, and their speed.   DM_PRECS=f16x3,f16i8,f32  DM_N=windows per case"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import _lib, model, synth
from oracle import oracle_np
precs = os.environ.get('DM_PRECS', 'f16x3,f16i8,f32').split(',')
n = int(os.environ.get('DM_N', '20000'))
x = synth.synthetic_windows(n, seed=3)
print("max |dp| vs the fp32 C oracle / vs float64 (numpy) ; flips away from near ties;  %d windows" % n)
for scale in (1.0, 4.0, 16.0):
    for seed in (7, 26, 21):
        w = synth.synthetic_weights(seed, scale)
        ref, ref_cls = oracle_np.predict_windows_c(w, x)
        p64 = oracle_np.predict_windows_np(w, x, np.float64)[0]
        near = np.abs(ref[:, 1] - 0.5) < 1e-4
        line = "scale %4g seed %2d  oracle vs f64 %.2e |" % (scale, seed, np.abs(ref - p64).max())
        m = model.BiLSTMModel(w, 0)
        for name in precs:
            m.set_precision(name)
            p, c = m.predict_windows(x)
            line += "  %s %.2e / %.2e F%d" % (name, np.abs(p - ref).max(), np.abs(p - p64).max(), int(((c != ref_cls) & ~near).sum()))
        m.close()
        print(line, flush=True)
w = synth.synthetic_weights(26, 4.0)
N = 65536
xx = synth.synthetic_windows(N, seed=1)
for rnd in range(2):
    for name in precs:
        m = model.BiLSTMModel(w, 0)
        m.set_precision(name)
        m.set_option(_lib.DM_OPT_PROFILE, 1)
        dx = model.DeviceArray.from_host(xx, 0); dc = model.DeviceArray((N,), np.uint8, 0)
        m.predict_windows(dx, cls=dc, want_prob=False)
        m.profile_reset()
        for _ in range(30): m.predict_windows(dx, cls=dc, want_prob=False)
        ms, launches, _ = m.profile_get()
        print("%-6s %.3f ms per 65,536 windows -> %.3g windows/s" % (name, ms / launches, N / (ms / launches) * 1e3), flush=True)
        m.close()
