import time, sys, os
T0 = time.time()
sys.path.insert(0, os.getcwd())
def lap(what, t=[T0]):
    now = time.time(); print('%-28s %.3f' % (what, now - t[0]), flush=True); t[0] = now
os.makedirs('/tmp/st/real', exist_ok=True); os.makedirs('/tmp/st/in', exist_ok=True)
for i in range(2016):
    p = '/tmp/st/real/raw_%04d.dmraw.npz' % i
    if not os.path.exists(p):
        open(p, 'wb').write(b'x' * 1000)
    for k in range(6):
        l = '/tmp/st/in/rep%d_raw_%04d.dmraw.npz' % (k, i)
        if not os.path.lexists(l):
            os.symlink(p, l)
lap('make links')
from deepmod_amd import detect
lap('import detect')
files = detect.discover_inputs('/tmp/st/in', False); lap('discover')
items = detect.plan_batches_sized(files, 22, int(0.8 * (128 << 20))); lap('plan')
from deepmod_amd import stream; lap('import stream')
import multiprocessing
ctx = multiprocessing.get_context('spawn')
w = stream.WorkList(items, ctx); lap('worklist')
q = ctx.Queue(); lap('ctx.Queue')
print(stream.usable_cpus()); lap('usable_cpus')
from deepmod_amd import comm, signal, _lib; lap('import comm signal')
_lib.load(); lap('lib load')
from deepmod_amd import readmap, synth
e = stream.StreamEngine({'outFolder': '/tmp/st/o', 'Base': 'C'}, None, 0, 1); lap('engine')
prefix = '/tmp/st/model/m'; os.makedirs('/tmp/st/model', exist_ok=True)
synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0); lap('write ckpt')
mo = {'outFolder': '/tmp/st/o', 'Base': 'C', 'modfile': [prefix, '/tmp/st/model/'], 'windowsize': 21}
try:
    b = stream.HipBackend(mo, 0); lap('HipBackend')
    b.close(); lap('close')
except Exception as exc:
    print('backend:', repr(exc)[:300])
