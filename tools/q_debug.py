import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import model, synth
from oracle import oracle_np
from deepmod_amd import _lib
if os.environ.get('DM_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['DM_LIB'])
base = synth.synthetic_weights(26, 1.0)
x = synth.synthetic_windows(64, seed=5)
mode = sys.argv[1] if len(sys.argv) > 1 else 'bias'
w0 = {k: v.copy() for k, v in base.items()}
if mode == 'bias':
    for k in w0:
        if k.endswith('kernel'):
            w0[k][:] = 0
bad = {}
for d, off in (('fw', 0), ('bw', 100)):
    errs = []
    for u in range(100):
        w = {k: v.copy() for k, v in w0.items()}
        w['Variable'][:] = 0
        w['Variable'][off + u, 1] = 4.0
        w['Variable_1'][:] = 0
        ref, _ = oracle_np.predict_windows_c(w, x)
        os.environ['DM_F16X3_SHAPE'] = '16'
        m = model.BiLSTMModel(w, 0, precision='f16x3')
        p, _ = m.predict_windows(x)
        m.close()
        e = np.abs(p - ref).max(axis=1)
        errs.append(float(e.max()))
        if e.max() > 1e-5:
            bad[(d, u)] = (float(e.max()), np.flatnonzero(e > 1e-5)[:8].tolist())
    print(d, "units with error > 1e-5:", [u for u in range(100) if errs[u] > 1e-5], flush=True)
for k, v in list(bad.items())[:12]:
    print(k, v)
