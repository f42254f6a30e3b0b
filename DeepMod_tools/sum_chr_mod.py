#!/usr/bin/env python
"""Same command line as the reference's DeepMod_tools/sum_chr_mod.py:
    python sum_chr_mod.py pred_folder-of-DeepMod Base-of-interest unique-fileid-in-sum-file [chr-list]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmod_amd import merge  # noqa: E402

if len(sys.argv) < 4:
    print("Usage: python {} pred_folder-of-DeepMod Base-of-interest unique-fileid-in-sum-file [chr-list]".format(sys.argv[0]))
    print("       pred_folder-of-DeepMod: the prediction must in its sub-folder.")
    sys.exit(1)
merge.sum_chr_mod(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4].split(',') if len(sys.argv) > 4 else None)
