"""Golden for the motif-position generator: runs the REFERENCE's own tool
/root/reference/DeepMod_tools/generate_motif_pos.py (plain Python) on a small synthetic FASTA (soft-masked stretches,
N runs, CpGs at both sequence ends) and stores input + outputs as text in motif_case.json.

Run only here (needs /root/reference):  python tests/golden/make_golden_motif.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(9)
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, 'g.fa')
seqs = {'chr1': 'CG' + ''.join(rng.choice(list('ACGTacgtN'), 3000, p=[.2, .22, .22, .2, .03, .04, .04, .03, .02])) + 'NNNNCGCGC',
        'chr2': ''.join(rng.choice(list('ACGT'), 1500)) + 'C'}
with open(fa, 'w') as fh:
    for k, v in seqs.items():
        fh.write('>%s description text\n' % k)
        for i in range(0, len(v), 70):
            fh.write(v[i:i + 70] + '\n')
subprocess.check_call([sys.executable, '/root/reference/DeepMod_tools/generate_motif_pos.py', fa, tmp + '/out', 'C', 'CG', '0', '1,2'],
                      stdout=subprocess.DEVNULL)
outputs = {fn: open(os.path.join(tmp, 'out', fn)).read() for fn in sorted(os.listdir(tmp + '/out'))}
for fn, t in outputs.items():
    print(fn, len(t.splitlines()))
json.dump({'fasta': open(fa).read(), 'argv': ['C', 'CG', '0', '1,2'], 'outputs': outputs}, open(os.path.join(HERE, 'motif_case.json'), 'w'))
