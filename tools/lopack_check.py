"""Dev tool (GPU box, round 6): VERDICT r05 item 3a - the lo halves of the default kernel's new h packed by v_fma_mixlo / mixhi_f16 (tools/ablate.py variants
q_lopack_mix: real, q_lopack_none: timing only).  Timing of the three builds alternating, and whether q_lopack_mix is bit-identical to the base.
    python tools/ablate.py build q_base,q_lopack_mix,q_lopack_none      (build container)
    python tools/lopack_check.py                                          (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.path.join(ROOT, "tools", "_abl")
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from deepmod_amd import _lib, model, synth
_lib.LIB_PATH = sys.argv[1]
w = synth.synthetic_weights(26, 4.0)
x = synth.synthetic_windows(70001, seed=9)
m = model.BiLSTMModel(w, 0)
prob, cls = m.predict_windows(x)
np.save(sys.argv[2], prob)
m.set_option(_lib.DM_OPT_PROFILE, 1)
dx = model.DeviceArray.from_host(synth.synthetic_windows(65536, seed=1), 0)
dc = model.DeviceArray((65536,), np.uint8, 0)
for _ in range(40): m.predict_windows(dx, cls=dc, want_prob=False)
m.sync(); m.profile_reset()
for _ in range(200): m.predict_windows(dx, cls=dc, want_prob=False)
ms, n, _ = m.profile_get()
print("%%.4f" %% (ms / n))
''' % ROOT
names = ["q_base", "q_lopack_mix", "q_lopack_none"]
times = {n: [] for n in names}
for rep in range(4):
    for n in names:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ABL, "lib_%s.so" % n), "/tmp/lopack_%s.npy" % n], capture_output=True, text=True)
        if out.returncode:
            print(n, "FAILED", out.stderr[-500:])
            continue
        times[n].append(float(out.stdout.strip().splitlines()[-1]))
import numpy as np
base = np.load("/tmp/lopack_q_base.npy")
for n in names:
    p = np.load("/tmp/lopack_%s.npy" % n)
    t = times[n]
    print("%-14s ms per 65,536-window launch (200 launches after 40 untimed, 4 alternating passes): %s  median %.4f  | vs base: bit-identical %s, max |dp| %.3g"
          % (n, " ".join("%.4f" % v for v in t), sorted(t)[len(t) // 2], bool(np.array_equal(p.view(np.uint32), base.view(np.uint32))), float(np.abs(p - base).max())))
b = sorted(times["q_base"])[len(times["q_base"]) // 2]
for n in names[1:]:
    print("%s / base = %.4f" % (n, sorted(times[n])[len(times[n]) // 2] / b))
