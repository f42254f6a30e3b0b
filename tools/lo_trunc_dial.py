"""Dev tool (GPU box): the operand-toggle dial of VERDICT r03 item 1a.

For every setting (m_w = low mantissa bits rounded away from the weights' lo halves, host packer knob DM_WLO_TRUNC; m_a = low mantissa bits
zeroed in the activations' lo halves, kernel build tools/_abl/lib_alo<m>.so) it prints
  * the error tail of the default kernel on 10^6 windows against the fp32 C oracle (two weight seeds at scale 4), and
  * ms per 65,536-window launch, socket power, shader clock and J per launch over a few seconds of back-to-back launches.
One process per setting (the library and the packer knob are fixed at load / model creation); the oracle output is computed once per
weight seed and cached in gpurun_out/.

    python tools/lo_trunc_dial.py [settings]        settings like 0:0,3:0,5:0,5:5  (m_w:m_a)

Experiment builds (the product library does not read DM_WLO_TRUNC): tools/_abl/lib_alo<m>.so = hipcc ... -DDM_WLO_TRUNC_ENV -DDM16S_ALO_TRUNC=<m>
(m = 0: lib_alo0.so), see tools/ablate.py variants alo0 / alo3 / alo5 / alo6.
"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

N = int(os.environ.get('DM_N', '1000000'))
SEEDS = ((17, 4.0), (26, 4.0))
CACHE = os.path.join(ROOT, 'gpurun_out', 'dial_cache')


def oracle_ref(seed, scale):
    from deepmod_amd import synth
    from oracle import oracle_np
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, 'ref_%d_%g_%d.npy' % (seed, scale, N))
    if os.path.exists(path):
        return np.load(path)
    x = synth.synthetic_windows(N, seed=20260928)
    w = synth.synthetic_weights(seed, scale)
    ref = np.concatenate([oracle_np.predict_windows_c(w, x[o:o + 65536])[0] for o in range(0, N, 65536)])
    np.save(path, ref)
    return ref


def child(mw, ma):
    from deepmod_amd import _lib, model, synth
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', '_abl', 'lib_alo%d.so' % ma)
    x = synth.synthetic_windows(N, seed=20260928)
    out = []
    for seed, scale in SEEDS:
        ref = oracle_ref(seed, scale)
        m = model.BiLSTMModel(synth.synthetic_weights(seed, scale), 0)
        p = np.concatenate([m.predict_windows(x[o:o + 65536])[0] for o in range(0, N, 65536)])
        m.close()
        d = np.abs(p - ref).max(axis=1)
        out.append("seed %d: max %.3g p99.99 %.3g median %.3g >3e-5: %d" % (seed, d.max(), np.quantile(d, 0.9999), np.median(d), int((d > 3e-5).sum())))
    print("m_w %d m_a %d | " % (mw, ma) + " | ".join(out), flush=True)


def main():
    settings = [tuple(int(v) for v in s.split(':')) for s in (sys.argv[1] if len(sys.argv) > 1 else '0:0,3:0,5:0,7:0,0:3,0:5,5:5').split(',')]
    t0 = time.time()
    for seed, scale in SEEDS:
        oracle_ref(seed, scale)
    print("oracle: %.0f s" % (time.time() - t0), flush=True)
    for mw, ma in settings:
        env = dict(os.environ, DM_WLO_TRUNC=str(mw))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--child', str(mw), str(ma)], env=env)
        env['DM_LIB'] = os.path.join(ROOT, 'tools', '_abl', 'lib_alo%d.so' % ma)
        tmp = os.path.join(ROOT, 'gpurun_out', 'dial_power.txt')
        r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'power_trace.sh'), tmp, sys.executable, os.path.join(ROOT, 'tools', 'bench_loop.py'), 'f16x3', '5'],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        ms = w = clk = None
        for l in r.stdout.splitlines():
            if 'ms per' in l:
                ms = float(l.split('launches,')[1].split('ms')[0])
            if l.startswith('socket power'):
                w = float(l.split('busy median')[1].split(',')[0])
            if l.startswith('sclk'):
                clk = float(l.split('median')[1].split(',')[0])
        if ms and w:
            print("m_w %d m_a %d | %.4f ms per launch, %.0f W, %.0f MHz, %.3f J per launch" % (mw, ma, ms, w, clk, ms * 1e-3 * w), flush=True)
        else:
            print("m_w %d m_a %d | power trace failed:\n%s" % (mw, ma, r.stdout[-600:]), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
