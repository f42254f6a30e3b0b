"""Dev tool (GPU box): where the host time of the streaming detect goes.  One worker batch of synthetic reads is prepared
(deepmod_amd/stream.py:prepare_batch) under cProfile, then submitted to the device, so that host-side hot spots and the
rows/s of one feeder thread are visible.
    python tools/e2e_rate.py raw|feat|packed [n_reads]"""
import cProfile, io, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepmod_amd import predstore, signal as dmsignal, stream, synth, synth_reads

kind = sys.argv[1] if len(sys.argv) > 1 else "raw"
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tmp = tempfile.mkdtemp()
t0 = time.time()
fasta = None
if kind == "feat":
    files = synth_reads.write_synthetic_run(tmp + "/in", n_reads=n_reads, reads_per_file=10, genome_len=200000, seed=3, chrom="chrS")
elif kind == "packed":
    files = synth_reads.write_synthetic_packed_run(tmp + "/in", genome_len=200000, coverage=n_reads * 6000 / 200000.0, reads_per_file=10,
                                                   seed=3, chrom="chrS")
else:
    files, fasta = synth_reads.write_synthetic_raw_run(tmp + "/in", n_reads=n_reads, reads_per_file=10, genome_len=200000, seed=3,
                                                       chrom="chrS", min_len=2000, max_len=8000)
print("generated %d files in %.1f s" % (len(files), time.time() - t0), flush=True)
prefix = tmp + "/model/m"
os.makedirs(tmp + "/model")
synth.write_synthetic_checkpoint(prefix, seed=26, scale=4.0)
mo = {'fnum': 7, 'hidden': 100, 'windowsize': 21, 'modfile': [prefix, tmp + "/model/"], 'outFolder': tmp + "/out", 'FileID': 'r', 'wrkBase': tmp + "/in",
      'outLevel': 3, 'Ref': fasta, 'alignStr': 'minimap2', 'region': [[None, None, None]], 'ConUnk': True, 'SignalGroup': 'simple', 'Base': 'C'}
os.makedirs(mo['outFolder'], exist_ok=True)
backend = stream.HipBackend(mo, 0)
eng = stream.StreamEngine(mo, backend)
norm = dmsignal.SignalNormalizer(0)
make_norm = lambda: norm
eng.consume(stream.prepare_batch(mo, files[:1], make_norm))          # warm up (library load, first launches)
backend.sync()
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
pb = stream.prepare_batch(mo, files, make_norm)
pr.disable()
t_prep = time.time() - t0
t0 = time.time()
eng.consume(pb)
backend.sync()
t_dev = time.time() - t0
print("%s containers: %d reads, %d rows (%d windows): prepare %.3f s = %.3g rows/s per feeder thread (%s); submit + device %.3f s" %
      (kind, pb.n_reads, pb.n_rows, pb.n_windows, t_prep, pb.n_rows / t_prep, ", ".join("%s %.0f%%" % (k, 100 * v / t_prep) for k, v in pb.timing.items()),
       t_dev), flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(26)
print('\n'.join(s.getvalue().splitlines()[4:44]))
if os.environ.get('DM_PROFILE_CALLERS'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_callers(os.environ['DM_PROFILE_CALLERS'])
    print(s.getvalue()[:3000])
