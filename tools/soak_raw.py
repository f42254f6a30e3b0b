"""Dev tool (GPU box): the streaming CLI on RAW containers over and over - feeder counts 1..6, feature rows assembled on the device (round 5
default) and on the host (DEEPMOD_ROWS_ON_DEVICE=0): every run must exit 0 and write the same BED bytes (a race in the signal server, the
device-form hand-over or the assemble launch would show as a differing digest).
    python tools/soak_raw.py [n_reads] [runs]"""
import hashlib, multiprocessing, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from deepmod_amd import synth
import e2e_detect_raw as E

if __name__ == "__main__":
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    ncpu = min(32, len(os.sched_getaffinity(0)))
    tmp = tempfile.mkdtemp()
    wrk = tmp + "/in"
    per = -(-n_reads // ncpu)
    with multiprocessing.get_context("spawn").Pool(ncpu) as pool:
        files = sum(pool.map(E._gen, [(wrk, p, per) for p in range(ncpu)]), [])
    prefix = tmp + "/model/m"
    os.makedirs(tmp + "/model")
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    rng = np.random.default_rng(2)
    digests, walls = {}, []
    for r in range(runs):
        threads = int(rng.integers(1, 7))
        on_dev = r % 2 == 0
        env = dict(os.environ, DEEPMOD_ROWS_ON_DEVICE="1" if on_dev else "0")
        out = "%s/out%d" % (tmp, r)
        cmd = [sys.executable, os.path.join(ROOT, "bin", "DeepMod.py"), "detect", "--wrkBase", wrk, "--Ref", wrk + "/genome.fa", "--modfile", prefix, "--outFolder", out,
               "--Base", "C", "--gpus", "1", "--threads", str(threads), "--FileID", "raw", "--alignStr", "minimap2"]
        t0 = time.time()
        res = subprocess.run(cmd, capture_output=True, text=True, env=env)
        walls.append(time.time() - t0)
        assert res.returncode == 0, (r, threads, on_dev, res.stdout[-1500:], res.stderr[-2500:])
        h = hashlib.sha256()
        for strand in "+-":
            h.update(open("%s/raw/mod_pos.chrS%s.C.bed" % (out, strand), "rb").read())
        digests.setdefault(h.hexdigest()[:16], []).append((threads, on_dev))
    print("%d runs on %d raw containers (1..6 feeders, rows on the device / on the host alternating): exit 0 every time, BED digests %s; wall %.2f..%.2f s"
          % (runs, len(files), {k: len(v) for k, v in digests.items()}, min(walls), max(walls)))
    assert len(digests) == 1, digests
