import time, os, sys
t0 = time.perf_counter()
sys.path.insert(0, os.getcwd())
import numpy as np
t1 = time.perf_counter()
from deepmod_amd import _lib, model as dm, synth
t2 = time.perf_counter()
lib = _lib.load()
t3 = time.perf_counter()
n = lib.dm_device_count()
t4 = time.perf_counter()
prefix = "/tmp/mm/m"
os.makedirs("/tmp/mm", exist_ok=True)
synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
t5 = time.perf_counter()
mo = {'fnum': 7, 'hidden': 100, 'windowsize': 21, 'outputlayer': ''}
dm.mCreateSession(7, 100, 21, mo)
sess = dm.new_session(0)
t6 = time.perf_counter()
dm.import_meta_graph(prefix + '.meta').restore(sess, dm.latest_checkpoint("/tmp/mm/") or prefix)
t7 = time.perf_counter()
m = sess.model
x = synth.synthetic_windows(4096, seed=1)
t8 = time.perf_counter()
m.predict_windows(x)
t9 = time.perf_counter()
m.predict_windows(x)
t10 = time.perf_counter()
print("numpy %.3f  package %.3f  dlopen %.3f  device_count(hipInit) %.3f  session %.3f  restore(model create, pack, upload) %.3f  first launch %.3f  second %.3f"
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t6 - t5, t7 - t6, t9 - t8, t10 - t9))
