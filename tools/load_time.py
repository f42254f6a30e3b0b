import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from deepmod_amd import model, synth, _lib
w = synth.synthetic_weights(26, 4.0)
m0 = model.BiLSTMModel(w, 0); m0.predict_windows(synth.synthetic_windows(256, seed=1)); m0.close()   # HIP context + first-launch costs out of the way
for rep in range(3):
    t0 = time.perf_counter(); m = model.BiLSTMModel(w, 0); t1 = time.perf_counter()
    m.predict_windows(synth.synthetic_windows(256, seed=1)); t2 = time.perf_counter()
    err, sel = m.calibrate_i8(); t3 = time.perf_counter()
    err2, sel2 = m.calibrate_i8(); t4 = time.perf_counter()
    print("create %.1f ms, first predict (pack + launch) %.1f ms, calibrate (first: packs int8 + fp32) %.1f ms, calibrate again %.1f ms; err %.3g sel %s" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, err, sel))
    m.close()
