"""Dev tool (GPU box): the TAIL of the probability error on 10^6 windows (BASELINE configs[1]) for every kernel, against the fp32 C oracle.
A maximum over 20,000 windows (tools/i8_check.py) says little about the worst window in a million."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import model, synth
from oracle import oracle_np
N = int(os.environ.get('DM_N', '1000000'))
x = synth.synthetic_windows(N, seed=20260928)
def trained_like():
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'trained_like_weights.npz'))
    return {k.replace('|', '/'): np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}


for seed, scale in (('trained-like', 0.0), (17, 4.0), (26, 4.0), (17, 1.0)):
    w = trained_like() if seed == 'trained-like' else synth.synthetic_weights(seed, scale)
    t0 = time.time()
    ref = np.concatenate([oracle_np.predict_windows_c(w, x[o:o + 65536])[0] for o in range(0, N, 65536)])
    m = model.BiLSTMModel(w, 0)
    line = "weights seed %s scale %g, %d windows (oracle %.0f s), class-1 fraction %.3f:" % (seed, scale, N, time.time() - t0, float((ref[:, 1] > 0.5).mean()))
    for prec in ('f16x3', 'f16i8', 'f32'):
        m.set_precision(prec)
        p = np.concatenate([m.predict_windows(x[o:o + 65536])[0] for o in range(0, N, 65536)])
        d = np.abs(p - ref).max(axis=1)
        line += "\n   %-6s max %.3g  p99.99 %.3g  p99.9 %.3g  median %.3g  windows > 5e-5: %d  > 1e-4: %d" % (
            prec, d.max(), np.quantile(d, 0.9999), np.quantile(d, 0.999), np.median(d), int((d > 5e-5).sum()), int((d > 1e-4).sum()))
    m.close()
    print(line, flush=True)
