"""Dev tool (GPU box): error of the classifier kernels against the C oracle as a function of the event length (feature 6) - nominal windows of
the trained-like model whose events in a chosen window position get length L."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import _lib, model, synth
from oracle import oracle_np
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'trained_like_weights.npz'))
w = {k.replace('|', '/'): np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}
ms = {'f32': model.BiLSTMModel(w, 0, precision='f32'), 'f16x3': model.BiLSTMModel(w, 0, precision='f16x3'), 'f16i8': model.BiLSTMModel(w, 0, precision='f16i8')}
m32 = model.BiLSTMModel(w, 0, precision='f16x3')
if m32.get_info(_lib.DM_INFO_HAS_F16S):      # experiment builds only (DM_WITH_F16S=1): the 32x32x16 kernel of rounds 2-3 beside the product kernel
    m32.set_option(_lib.DM_OPT_F16X3_SHAPE, 32); ms['f16x3/32'] = m32
n = 8192
for where in ('centre', 'row 3', 'all rows'):
    for L in (10, 100, 1000, 10000, 30000, 60000):
        x = synth.synthetic_windows(n, seed=5)
        if where == 'centre': x[:, 10, 6] = L
        elif where == 'row 3': x[:, 3, 6] = L
        else: x[:, :, 6] = L
        ref, rc = oracle_np.predict_windows_c(w, x)
        line = "%-9s L %6d: class-1 %.3f |" % (where, L, rc.mean())
        for k, m in ms.items():
            p, c = m.predict_windows(x)
            line += "  %s %.3g" % (k, np.abs(p - ref).max())
        print(line, flush=True)
