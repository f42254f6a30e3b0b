"""Dev tool (GPU box): A/B of the two kernels behind DM_PREC_F16X3 - lstm16s::bilstm_f16s_kernel<0> (32x32x16 MFMAs) and
lstm16q::bilstm_f16q_kernel (16x16x32 MFMAs), picked with DM_F16X3_SHAPE = 32 | 16 at model creation: error against the fp32 C oracle on
ragged sizes and several weight sets, then ms per 65,536-window launch (alternating)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import _lib, model, synth
from oracle import oracle_np
if os.environ.get('DM_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['DM_LIB'])


def make(w, shape):
    os.environ['DM_F16X3_SHAPE'] = str(shape)
    return model.BiLSTMModel(w, 0, precision='f16x3')


def trained_like():
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'trained_like_weights.npz'))
    return {k.replace('|', '/'): np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}


worst = {32: 0.0, 16: 0.0}
for name, w in (('scale 1', synth.synthetic_weights(21, 1.0)), ('scale 4', synth.synthetic_weights(26, 4.0)), ('trained-like', trained_like())):
    ms = {sh: make(w, sh) for sh in (32, 16)}
    for n in (1, 15, 16, 17, 31, 33, 128, 129, 1000, 4097, 20000):
        x = synth.synthetic_windows(n, seed=100 + n)
        ref, rcls = oracle_np.predict_windows_c(w, x)
        line = "%-12s n %6d:" % (name, n)
        for sh in (32, 16):
            p, c = ms[sh].predict_windows(x)
            e = float(np.abs(p - ref).max())
            near = np.abs(ref[:, 1] - 0.5) < 1e-4
            flips = int(((c.astype(np.int64) != rcls) & ~near).sum())
            worst[sh] = max(worst[sh], e)
            line += "  shape %d: max|dp| %.3g flips %d" % (sh, e, flips)
        print(line, flush=True)
    for m in ms.values():
        m.close()
print("worst: 32x32x16 %.3g, 16x16x32 %.3g" % (worst[32], worst[16]), flush=True)
w = synth.synthetic_weights(26, 4.0)
n = 65536
dx = model.DeviceArray.from_host(synth.synthetic_windows(n, seed=1), 0)
dc = model.DeviceArray((n,), np.uint8, 0)
for rep in range(2):
    for sh in (32, 16):
        m = make(w, sh)
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
        print("shape %d: %.4f ms per 65,536-window launch (%d launches)" % (sh, t / launches, launches), flush=True)
        m.close()
