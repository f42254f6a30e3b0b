"""Dev tool (GPU box): A/B of the two implementations of DM_PREC_F16X3 - lstm16s::bilstm_f16s_kernel<0> (one wave per SIMD) and
lstm16r::bilstm_f16r_kernel (matrix / cell wave pairs): bit equality of the outputs on ragged sizes, then ms per 65,536-window launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import _lib, model, synth
_lib.LIB_PATH = os.path.abspath(os.environ.get('DM_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), '_abl', 'lib_roles.so')))


def make(w, roles):
    return model.BiLSTMModel(w, 0, precision='f16x3r' if roles else 'f16x3')      # f16x3r: a library built with -DDM_WITH_F16X3_ROLES


ok = True
for seed, scale in ((21, 1.0), (26, 4.0)):
    w = synth.synthetic_weights(seed, scale)
    ms, mr = make(w, False), make(w, True)
    for n in (1, 31, 128, 129, 1000, 4097, 65536, 140000):
        x = synth.synthetic_windows(n, seed=100 + n)
        ps, cs = ms.predict_windows(x)
        pr, cr = mr.predict_windows(x)
        same = np.array_equal(ps.view(np.uint32), pr.view(np.uint32)) and np.array_equal(cs, cr)
        ok &= same
        print("scale %g n %6d: %s  max|dp| %.3g" % (scale, n, "bit-identical" if same else "DIFFERENT", float(np.abs(ps - pr).max())), flush=True)
    ms.close(); mr.close()
print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)
w = synth.synthetic_weights(26, 4.0)
n = 65536
dx = model.DeviceArray.from_host(synth.synthetic_windows(n, seed=1), 0)
dc = model.DeviceArray((n,), np.uint8, 0)
for rep in range(2):
    for roles in (False, True):
        m = make(w, roles)
        m.set_option(_lib.DM_OPT_PROFILE, 1)
        m.set_option(_lib.DM_OPT_ASYNC, 1)
        for _ in range(40):
            m.predict_windows(dx, cls=dc, want_prob=False)
        m.sync(); m.profile_reset()
        t0 = time.time()
        while time.time() - t0 < 3.0:
            for _ in range(50):
                m.predict_windows(dx, cls=dc, want_prob=False)
            m.sync()
        t, launches, _ = m.profile_get()
        print("%s: %.4f ms per 65,536-window launch (%d launches)" % ("roles (f16r)" if roles else "single (f16s)", t / launches, launches), flush=True)
        m.close()
