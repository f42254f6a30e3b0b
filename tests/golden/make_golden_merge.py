"""Golden for the per-chromosome BED merge: runs the REFERENCE's own tool
/root/reference/DeepMod_tools/sum_chr_mod.py (plain Python, no third-party imports) on small synthetic per-run BED
files laid out as DeepMod writes them, and stores inputs + outputs as text in merge_case.json.

Run only here (needs /root/reference):  python tests/golden/make_golden_merge.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def bed_text(rng, chrom, strand, base, n, lo, hi):
    pos = np.sort(rng.choice(np.arange(lo, hi), n, replace=False))
    lines = []
    for p in pos:
        cov = int(rng.integers(0, 40)) if rng.random() > 0.03 else int(rng.integers(900, 1500))
        mod = int(rng.integers(0, cov + 1)) if cov and rng.random() > 0.4 else 0
        # the detect stage's dialect: single spaces (myDetect.py:1116-1120)
        lines.append(' '.join([chrom, str(p), str(p + 1), base, str(min(cov, 1000)), strand, str(p), str(p + 1), '0,0,0', str(cov),
                               ('%d' % (100 * mod / cov if cov > 0 else 0)), str(mod)]) + '\n')
    return ''.join(lines)


def main():
    rng = np.random.default_rng(5)
    tmp = tempfile.mkdtemp()
    inputs = {}
    layout = [('runA/mod_pos.chr1+.C.bed', 'chr1', '+', 300), ('runA/mod_pos.chr1-.C.bed', 'chr1', '-', 280),
              ('runB/sub/mod_pos.chr1+.C.bed', 'chr1', '+', 320), ('runB/sub/mod_pos.chr1-.C.bed', 'chr1', '-', 200),
              ('runC/x/y/mod_pos.chr1+.C.bed', 'chr1', '+', 150),
              ('runA/mod_pos.chr2+.C.bed', 'chr2', '+', 100), ('runB/sub/mod_pos.chr2-.C.bed', 'chr2', '-', 120),
              ('runA/mod_pos.chrM+.C.bed', 'chrM', '+', 50)]
    for rel, chrom, strand, n in layout:
        text = bed_text(rng, chrom, strand, 'C', n, 1000, 1600)
        inputs[rel] = text
        os.makedirs(os.path.dirname(os.path.join(tmp, rel)), exist_ok=True)
        open(os.path.join(tmp, rel), 'w').write(text)
    subprocess.check_call([sys.executable, '/root/reference/DeepMod_tools/sum_chr_mod.py', tmp, 'C', 'merged', 'chr1,chr2,chr7'],
                          stdout=subprocess.DEVNULL)
    outputs = {fn: open(os.path.join(tmp, fn)).read() for fn in sorted(os.listdir(tmp)) if fn.endswith('.bed')}
    for fn, t in outputs.items():
        print(fn, len(t.splitlines()), 'lines')
    json.dump({'inputs': inputs, 'outputs': outputs, 'argv': ['C', 'merged', 'chr1,chr2,chr7']},
              open(os.path.join(HERE, 'merge_case.json'), 'w'))


if __name__ == '__main__':
    main()
