"""Dev tool (GPU box): the streaming CLI run over and over on one input with varying feeder counts - every run must end with
exit code 0 and the same BED bytes (races in the hand-over, the work list, the slots or the copy stream would show here).
    python tools/soak_cli.py [runs] [coverage]"""
import hashlib, multiprocessing, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from deepmod_amd import synth
import e2e_detect_packed as E

if __name__ == "__main__":
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cov = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    tmp = tempfile.mkdtemp()
    wrk = tmp + "/reads"
    total_files = int(np.ceil(cov * E.GENOME_LEN / 6000.0 / E.READS_PER_FILE))
    ncpu = min(32, len(os.sched_getaffinity(0)))
    chunk = int(np.ceil(total_files / ncpu))
    with multiprocessing.get_context("spawn").Pool(ncpu) as pool:
        files = sum(pool.map(E._gen, [(wrk, i, chunk, cov) for i in range(0, total_files, chunk)]), [])
    prefix = tmp + "/model/m"
    os.makedirs(tmp + "/model")
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    rng = np.random.default_rng(1)
    digests, walls = set(), []
    for r in range(runs):
        nf = int(rng.integers(1, 7))
        extra = [[], ["--files_per_thread", "2"], ["--files_per_thread", "9"]][int(rng.integers(0, 3))]
        env = dict(os.environ)
        if rng.random() < 0.3:
            env["DEEPMOD_SELECT_BASE"] = "0"              # every window classified: the classic hand-over form
        out = "%s/out%d" % (tmp, r)
        t0 = time.time()
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "DeepMod.py"), "detect", "--wrkBase", wrk, "--modfile", prefix, "--outFolder", out,
                              "--Base", "C", "--gpus", "1", "--threads", str(nf), "--FileID", "s"] + extra, capture_output=True, text=True, env=env, timeout=600)
        walls.append(time.time() - t0)
        if res.returncode:
            print("run %d (%d feeders, %s) FAILED rc=%d\n%s\n%s" % (r, nf, extra, res.returncode, res.stdout[-1500:], res.stderr[-3000:]))
            sys.exit(1)
        h = hashlib.sha256()
        for f in sorted(os.listdir(out + "/s")):
            if f.endswith(".bed"):
                h.update(f.encode() + open(out + "/s/" + f, "rb").read())
        digests.add(h.hexdigest())
        leftovers = [f for f in os.listdir("/dev/shm") if f.startswith("deepmod")]
        if len(digests) != 1 or leftovers:
            print("run %d (%d feeders, %s): BED differs from the runs before it, or shared memory left behind: %s" % (r, nf, extra, leftovers))
            sys.exit(1)
    print("%d runs (%d containers, 1..6 feeders, three batch sizes, both hand-over forms): exit 0 every time, one BED digest %s, nothing left in /dev/shm; wall %.2f..%.2f s"
          % (runs, len(files), next(iter(digests))[:16], min(walls), max(walls)))
