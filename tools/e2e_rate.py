"""Dev tool (GPU box): wall-clock of the host pipeline around the classifier for synthetic reads, with a cProfile of
one worker batch (mDetect1) so that host-side hot spots are visible."""
import cProfile, io, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict
import numpy as np
from deepmod_amd import detect, model as dm, predstore, synth, synth_reads

kind = sys.argv[1] if len(sys.argv) > 1 else "feat"
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tmp = tempfile.mkdtemp()
t0 = time.time()
if kind == "feat":
    files = synth_reads.write_synthetic_run(tmp + "/in", n_reads=n_reads, reads_per_file=10, genome_len=200000, seed=3, chrom="chrS")
    fasta = None
else:
    files, fasta = synth_reads.write_synthetic_raw_run(tmp + "/in", n_reads=n_reads, reads_per_file=10, genome_len=200000, seed=3,
                                                       chrom="chrS", min_len=2000, max_len=8000)
print("generated %d files in %.1f s" % (len(files), time.time() - t0), flush=True)
prefix = tmp + "/model/m"
os.makedirs(tmp + "/model")
synth.write_synthetic_checkpoint(prefix, seed=9, scale=4.0)
mo = {'fnum': 7, 'hidden': 100, 'windowsize': 21, 'modfile': [prefix, tmp + "/model/"], 'outFolder': tmp + "/out/", 'FileID': 'r', 'wrkBase': tmp + "/in",
      'outLevel': 3, 'Ref': fasta, 'alignStr': 'minimap2', 'region': [[None, None, None]], 'ConUnk': True, 'SignalGroup': 'simple', 'Base': 'C'}
_, init_l, _, _, _, X, Y, _, _, _, _, mfpred = dm.mCreateSession(7, 100, 21, mo)
sess = dm.new_session(0)
dm.import_meta_graph(prefix + '.meta').restore(sess, prefix)
sp = defaultdict()
os.makedirs(tmp + "/out/r/0", exist_ok=True)
sp.update({'ctfolder': tmp + "/out/r/0", 'batchid': 0, 'Mod': [], 'Error': defaultdict(list), 'rnn': (sess, X, Y, init_l, mfpred)})
detect.mDetect1(mo, sp, files[:1])                      # warm up (library load, first launches)
sp['Mod'] = []
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
detect.mDetect1(mo, sp, files)
pr.disable()
dt = time.time() - t0
bases = sum(int(l.split()[0] != '') for l in [])  # placeholder
nb = 0
for key_file in [sp['ctfolder'] + '/rnn.pred.detail.npz.0']:
    z = np.load(key_file, allow_pickle=False)
    nb = sum(len(z[k]) for k in z.files if k.endswith('/refbasei'))
print("%s containers: %d reads, %d table rows in %.2f s -> %.3g rows/s (errors: %s)" % (kind, len(sp['Mod']), nb, dt, nb / dt, dict(sp['Error'])), flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22)
print('\n'.join(s.getvalue().splitlines()[4:40]))
