import sys, ctypes, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmod_amd import _lib, model, summary, synth, signal, cluster
hip = ctypes.CDLL('/opt/rocm/lib/libamdhip64.so')
def free_mb():
    f=ctypes.c_size_t(); t=ctypes.c_size_t(); hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); return f.value/2**20
w = synth.synthetic_weights(1,1.0); x = synth.synthetic_windows(70000, seed=1)
raw = np.clip(np.round(np.random.default_rng(0).normal(480,70,200000)),0,2000).astype(np.int16)
st = np.arange(0,199000,10).astype(np.uint64); ln = np.full(len(st),10,np.uint64)
base=None
for it in range(25):
    m = model.BiLSTMModel(w,0); m.set_precision(("f32", "f16x3", "auto", "f16i8")[it % 4])      # "auto": the calibration gate's buffers come and go too
    p,c = m.predict_windows(x)
    s = summary.PositionSummary(1000000,0); s.add_classified(np.arange(70000,dtype=np.int64), np.full(70000,3,np.uint8), c, 70000); s.fetch(); s.close()
    nz = signal.SignalNormalizer(0); nz.event_stats(raw, st, ln); nz.close()
    m.close()
    if it in (2, 24): print("iteration", it, "free MB", round(free_mb()))
