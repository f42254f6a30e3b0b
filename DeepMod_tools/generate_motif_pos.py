#!/usr/bin/env python
"""Same command line as the reference's DeepMod_tools/generate_motif_pos.py:
    python generate_motif_pos.py ref.fa result-folder Base Motif Position-of-Base-in-Motif [chr-list without 'chr']"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmod_amd import motif  # noqa: E402

if len(sys.argv) < 6:
    print("Usage: python {} ref.fa result-folder Base Motif Position-of-Base-in-Motif [chr-list]".format(sys.argv[0]))
    sys.exit(1)
chrkeys = ["chr%s" % cid for cid in sys.argv[6].split(',')] if len(sys.argv) > 6 else None
motif.generate_motif_pos(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), chrkeys)
