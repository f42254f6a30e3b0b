"""Dev tool: per-wave timeline of one stage of the tile-major split-f16 kernel (a -DDM16T_TRACE build).
    python tools/experiments/f16t/trace_f16t.py build     # here (cross-compile tools/_abl/lib_f16t_trace.so)
    python tools/experiments/f16t/trace_f16t.py run       # on the GPU box
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(ROOT, "tools", "_abl", "lib_f16t_trace%s.so" % os.environ.get("DM_TAG", ""))
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = os.path.join(ROOT, "deepmod_amd", "csrc", "deepmod_hip.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB, src, "-ldl",
                           "-DDM_EXPERIMENT_F16T", "-I" + HERE, "-DDM16T_TRACE=%s" % os.environ.get("DM_TRACE_LAYER", "1")] + sys.argv[2:], cwd=os.path.dirname(src))
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np
from deepmod_amd import _lib, model, synth
_lib.LIB_PATH = LIB
w = synth.synthetic_weights(26, 4.0)
n = 65536
x = synth.synthetic_windows(n, seed=1)
m = model.BiLSTMModel(w, 0, precision="f16x3t")
m.set_option(_lib.DM_OPT_PROFILE, 1)
dx = model.DeviceArray.from_host(x, 0); dc = model.DeviceArray((n,), np.uint8, 0)
for _ in range(3): m.predict_windows(dx, cls=dc, want_prob=False)
m.profile_reset()
for _ in range(10): m.predict_windows(dx, cls=dc, want_prob=False)
ms, launches, _ = m.profile_get()
print("f16x3t (trace build) %.3f ms per launch" % (ms / launches))
lib = _lib.load()
lib.dm_debug_timing.restype = ctypes.c_longlong
lib.dm_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
cnt = lib.dm_debug_timing(m._h, None, 0)
buf = np.zeros(cnt, np.uint64)
lib.dm_debug_timing(m._h, buf.ctypes.data, cnt)
t = buf[:4 * 13 * 4].reshape(4, 13, 4).astype(np.int64)
t0 = t[:, 0, 0].min()
print("stage of layer 1 (13 tiles x 39 MFMAs = 1,248 MFMA cycles each), cycles relative to the first wave entering the stage")
print("tile   " + "".join("   w%d:start  mfma  +wait  +barrier" % w_ for w_ in range(4)))
for T in range(13):
    print("%4d   " % T + "".join("  %9d %5d %6d %8d" % (t[w_, T, 0] - t0, t[w_, T, 1] - t[w_, T, 0], t[w_, T, 2] - t[w_, T, 1], t[w_, T, 3] - t[w_, T, 2]) for w_ in range(4)))
print("stage length per wave:", [int(t[w_, 12, 3] - t[w_, 0, 0]) for w_ in range(4)])
k = buf[4 * 13 * 4:4 * 13 * 4 + 64].reshape(4, 16).astype(np.int64)
print("tile 5, start of each k16-step relative to the tile start (3 MFMAs = 96 MFMA cycles per step):")
for w_ in range(4):
    print("  wave %d: " % w_ + " ".join("%5d" % (k[w_, i] - t[w_, 5, 0]) for i in range(13)) + "   steps: " + " ".join("%4d" % (k[w_, i + 1] - k[w_, i]) for i in range(12)))
