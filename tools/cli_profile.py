"""Dev tool (GPU box): where the MANAGER process of `bin/DeepMod.py detect` (streaming mode) spends its time - cProfile of the command on a
small config-3-like input (the ranks and feeders are other processes: their time shows here as waiting)."""
import multiprocessing, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from deepmod_amd import synth
import e2e_detect_packed as E
if __name__ == "__main__":
    cov = 30.0
    tmp = tempfile.mkdtemp()
    wrk = tmp + "/reads"
    total_files = int(np.ceil(cov * E.GENOME_LEN / 6000.0 / E.READS_PER_FILE))
    ncpu = min(32, len(os.sched_getaffinity(0)))
    chunk = int(np.ceil(total_files / ncpu))
    with multiprocessing.get_context("spawn").Pool(ncpu) as pool:
        sum(pool.map(E._gen, [(wrk, i, chunk, cov) for i in range(0, total_files, chunk)]), [])
    prefix = tmp + "/model/m"
    os.makedirs(tmp + "/model")
    synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
    cmd = [sys.executable, "-m", "cProfile", "-s", "cumtime", os.path.join(ROOT, "bin", "DeepMod.py"), "detect", "--wrkBase", wrk, "--modfile", prefix,
           "--outFolder", tmp + "/out", "--Base", "C", "--gpus", "1", "--threads", "2", "--FileID", "p"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    lines = r.stdout.splitlines()
    i = next(k for k, l in enumerate(lines) if "cumulative" in l or "cumtime" in l)
    print("\n".join(l[:190] for l in lines[max(0, i - 8):i + 45]))
